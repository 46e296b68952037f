// Micro-benchmark (development aid): what a time stamp costs on gfx950 -- s_memtime + s_waitcnt against
// s_getreg_b32 of hardware register 29 (SHADER_CYCLES on later architectures; does it tick here?).
// build: hipcc --offload-arch=gfx950 -O3 -o stamp stamp.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void __launch_bounds__(128) k_memtime(unsigned long long *out, int n)
{
    unsigned long long t0 = clock64(), acc = 0;
    for (int i = 0; i < n; i++) {
        REP64({ unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); acc += t; })
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}
__global__ void __launch_bounds__(128) k_getreg(unsigned long long *out, int n)
{
    unsigned long long t0 = clock64();
    unsigned acc = 0, first = 0, last = 0;
    asm volatile("s_getreg_b32 %0, hwreg(29)" : "=s"(first));
    for (int i = 0; i < n; i++) {
        REP64({ unsigned t; asm volatile("s_getreg_b32 %0, hwreg(29)" : "=s"(t)); acc += t; last = t; })
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = ((unsigned long long) first << 32) | last; }
    if (acc == 12345u) out[0] = acc;
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 64 * 16);
    std::vector<unsigned long long> h(128);
    int n = 100;
    for (int waves = 1; waves <= 2; waves++) {
        hipLaunchKernelGGL(k_memtime, dim3(1), dim3(64 * waves), 0, 0, d, n); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost);
        printf("s_memtime + s_waitcnt, %d wave(s): %.1f cycles per stamp\n", waves, (double) h[0] / (64.0 * n));
        hipLaunchKernelGGL(k_getreg, dim3(1), dim3(64 * waves), 0, 0, d, n); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost);
        printf("s_getreg_b32 hwreg(29), %d wave(s): %.1f cycles per stamp; first %u last %u (clock64 span %llu)\n", waves,
               (double) h[0] / (64.0 * n), (unsigned) (h[1] >> 32), (unsigned) h[1], h[0]);
    }
    return 0;
}
