// Micro-benchmark (development aid, round 2): single-wave issue rates and latencies on gfx950
// that the quantiser loop design depends on.
// build: hipcc --offload-arch=gfx950 -O3 -o lat2 lat2.hip   (generated layout; edit freely)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
__global__ void __launch_bounds__(64) k_valu_indep8(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed,b=1,c=2,d=3,e=4,f=5,g=6,h=7;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) : "v"(3u));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a+b+c+d+e+f+g+h)==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_valu_indep2(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed,b=1;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n" : "+v"(a),"+v"(b) : "v"(3u));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a+b)==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_valu_dep(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(a) : "v"(3u));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_f64_indep4(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    double a=seed,b=1,c=2,d=3;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n" : "+v"(a),"+v"(b),"+v"(c),"+v"(d) : "v"(1.5));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a+b+c+d)==12345.0) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_cvt_indep4(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    float a=seed,b=1,c=2,d=3; double x,y,z,w;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"=&v"(x),"=&v"(y),"=&v"(z),"=&v"(w));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a+b+c+d)==12345.0f) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_lds_dep(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=(threadIdx.x*4)&4095;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n" : "+v"(a));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_lds_indep4(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=(threadIdx.x*4)&4095,b=a^64,c=a^128,d=a^256;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n s_waitcnt lgkmcnt(0)\n" : "+v"(a),"+v"(b),"+v"(c),"+v"(d));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a+b+c+d)==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_bperm_dep(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=(threadIdx.x*4)&255;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n" : "+v"(a));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_dpp_full(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed+threadIdx.x; unsigned s=0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
   "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
   "v_readlane_b32 %1, %0, 63\n" : "+v"(a), "+s"(s));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_ballot_rt(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed+threadIdx.x; unsigned s=0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_ff1_i32_b64 %1, vcc\n v_add_u32 %0, %0, %1\n" : "+v"(a), "+s"(s) :: "vcc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_rfl_rt(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=seed+threadIdx.x; unsigned s=0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 3\n v_add_u32 %0, %0, %1\n" : "+v"(a), "+s"(s) :: "scc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_salu_dep(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned s=seed;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 5\n" : "+s"(s) :: "scc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_sbranch(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned s=seed; unsigned a=threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n v_add_u32 %1, %1, 1\n 1:\n s_cmp_lg_u32 %0, 77\n s_cbranch_scc1 2f\n v_add_u32 %1, %1, 1\n 2:\n" : "+s"(s), "+v"(a) :: "scc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_execbr(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=threadIdx.x; unsigned long long sv=0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_gt_u32 vcc, 32, %0\n s_and_saveexec_b64 %1, vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 0\n 1:\n s_or_b64 exec, exec, %1\n" : "+v"(a), "+s"(sv) :: "vcc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_vcmp_cnd(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=threadIdx.x, b=seed;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc\n" : "+v"(a) : "v"(b) : "vcc");) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_wait_only(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" : "+v"(a));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a==12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_dswrite_read(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned a=(threadIdx.x*4)&4095; unsigned b=seed;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) { REP16(asm volatile("ds_write_b32 %0, %1\n ds_read_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, 1\n" : "+v"(a), "+v"(b));) }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (b==12345u) out[0] = 1;
}

__global__ void __launch_bounds__(64) k_gload_dep(unsigned long long *out, int n, unsigned seed, const unsigned *buf) {
    unsigned a = (threadIdx.x * 4) & 1023;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = buf[a >> 2];
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 12345u) out[0] = 1;
}
__global__ void __launch_bounds__(64) k_sload_dep(unsigned long long *out, int n, unsigned seed, const unsigned *buf) {
    unsigned a = seed & 1023;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = buf[__builtin_amdgcn_readfirstlane(a) >> 2];
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 12345u) out[0] = 1;
}
// mixed dependent work (LDS + VALU + DPP + scalar branch): how throughput scales with waves per SIMD
__global__ void __launch_bounds__(64) k_mix(unsigned long long *out, int n, unsigned seed) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 7 + 13) & 1023;
    __syncthreads();
    unsigned a = threadIdx.x + seed, acc = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            unsigned x = lds[(a + acc) & 1023];
            x = x * 3 + 1; x ^= x >> 3; x += a; x ^= x << 2; x += 7; x ^= x >> 5;
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0xB1, 0xf, 0xf, true);
            x += (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x4E, 0xf, 0xf, true);
            unsigned s = __builtin_amdgcn_readfirstlane(x);
            if (s & 1) acc += s >> 4; else acc ^= s;
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345u) out[0] = 1;
}
template <typename F> static double run(F launch, int nb) {
    unsigned long long *d; hipMalloc(&d, nb * 8);
    launch(d); launch(d); hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double) x; hipFree(d); return s / nb;
}
#define RUN(k, per, wps) { int nb = 256 * 4 * (wps); int n = 200; double c = run([&](unsigned long long *d){ hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, n, 1u); }, nb); \
    printf("%-16s waves/SIMD %d: %8.2f cycles per unit (%d units per asm block)\n", #k, wps, c / n / 16 / (per), per); }
int main() {
    unsigned *buf; hipMalloc(&buf, 4096); { std::vector<unsigned> h(1024); for (int i = 0; i < 1024; i++) h[i] = ((i * 7 + 13) & 1023) * 4; hipMemcpy(buf, h.data(), 4096, hipMemcpyHostToDevice); }
    for (int w : {1, 2, 4, 8}) {
        RUN(k_valu_indep8, 8, w) RUN(k_valu_indep2, 4, w) RUN(k_valu_dep, 4, w) RUN(k_f64_indep4, 4, w) RUN(k_cvt_indep4, 8, w)
        RUN(k_lds_dep, 4, w) RUN(k_lds_indep4, 1, w) RUN(k_bperm_dep, 2, w) RUN(k_dpp_full, 1, w) RUN(k_ballot_rt, 1, w) RUN(k_rfl_rt, 1, w)
        RUN(k_salu_dep, 4, w) RUN(k_sbranch, 2, w) RUN(k_execbr, 1, w) RUN(k_vcmp_cnd, 4, w) RUN(k_wait_only, 4, w) RUN(k_dswrite_read, 1, w)
        { int nb = 256*4*w; int n = 200; double c = run([&](unsigned long long *d){ hipLaunchKernelGGL(k_mix, dim3(nb), dim3(64), 0, 0, d, n, 1u); }, nb); printf("k_mix            waves/SIMD %d: %8.2f cycles per element\n", w, c / n / 16); }
        { int nb = 256*4*w; int n = 200; double c = run([&](unsigned long long *d){ hipLaunchKernelGGL(k_gload_dep, dim3(nb), dim3(64), 0, 0, d, n, 1u, buf); }, nb); printf("k_gload_dep      waves/SIMD %d: %8.2f cycles per load\n", w, c / n / 16); }
        { int nb = 256*4*w; int n = 200; double c = run([&](unsigned long long *d){ hipLaunchKernelGGL(k_sload_dep, dim3(nb), dim3(64), 0, 0, d, n, 1u, buf); }, nb); printf("k_sload_dep      waves/SIMD %d: %8.2f cycles per load\n", w, c / n / 16); }
    }
    return 0;
}
