// Micro-benchmark (development aid): cost of taken / not-taken scalar branches, dependent VALU
// chains, DPP steps and LDS round trips on gfx950 with 1 or 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void __launch_bounds__(128) k_taken(unsigned long long *out, int n, int one)
{
    unsigned v = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("v_add_u32 %0, %0, %1\n s_cmp_eq_u32 %2, 1\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n 1:\n" : "+v"(v) : "v"(3u), "s"(one) : "scc");)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345u) out[0] = v;
}
__global__ void __launch_bounds__(128) k_nottaken(unsigned long long *out, int n, int one)
{
    unsigned v = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("v_add_u32 %0, %0, %1\n s_cmp_eq_u32 %2, 1\n s_cbranch_scc0 1f\n s_nop 0\n s_nop 0\n 1:\n" : "+v"(v) : "v"(3u), "s"(one) : "scc");)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345u) out[0] = v;
}
__global__ void __launch_bounds__(128) k_valu_dep(unsigned long long *out, int n, int one)
{
    unsigned v = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n" : "+v"(v) : "v"(3u));)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345u) out[0] = v;
}
__global__ void __launch_bounds__(128) k_salu(unsigned long long *out, int n, int one)
{
    unsigned v = threadIdx.x;
    int s = one;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("s_add_i32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n s_add_i32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n" : "+s"(s) :: "scc");)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345) out[0] = v;
}
__global__ void __launch_bounds__(128) k_lds_dep(unsigned long long *out, int n, int one)
{
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 128) lds[i] = (i * 7 + 13) & 4095;
    __syncthreads();
    unsigned v = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) v = lds[v];
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345u) out[0] = v;
}
__global__ void __launch_bounds__(128) k_dpp_reduce(unsigned long long *out, int n, int one)
{
    unsigned v = threadIdx.x, acc = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            unsigned x = v + acc;
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0xB1, 0xf, 0xf, true);
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x4E, 0xf, 0xf, true);
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x141, 0xf, 0xf, true);
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x140, 0xf, 0xf, true);
            acc = __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345u) out[0] = v;
}


__global__ void __launch_bounds__(128) k_f64_add(unsigned long long *out, int n, int one)
{
    double v = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(v) : "v"(1.5));)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345.0) out[0] = 1;
}
__global__ void __launch_bounds__(128) k_cvt(unsigned long long *out, int n, int one)
{
    float f = threadIdx.x;
    double d;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        REP64(asm volatile("v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1\n" : "+v"(f), "=v"(d));)
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (f == 12345.0f) out[0] = 1;
}

template <typename K> static void run(const char *name, K kern, int per_iter, int blocks_per_cu)
{
    int n = 200;
    int nb = 256 * blocks_per_cu;
    unsigned long long *d;
    hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(128), 0, 0, d, n, 1);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(128), 0, 0, d, n, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double) x;
    printf("%-14s waves/SIMD %d: %.1f cycles per element (%d per loop iteration)\n", name, blocks_per_cu / 2, s / nb / n / per_iter, per_iter);
    hipFree(d);
}

int main()
{
    for (int b : {2, 4}) {
        run("taken-branch", k_taken, 64, b);
        run("nottaken", k_nottaken, 64, b);
        run("valu-dep x4", k_valu_dep, 64, b);
        run("salu-dep x4", k_salu, 64, b);
        run("lds-dep", k_lds_dep, 64, b);
        run("dpp-reduce", k_dpp_reduce, 16, b);
        run("add_f64 x4", k_f64_add, 64, b);
        run("cvt x4", k_cvt, 64, b);
    }
    return 0;
}
