// Development check: is __fsqrt_rn(x) == (float) sqrt((double) x) for every float in [1, 4)
// (all mantissas, both exponent parities; scaling by 4^k is exact)?  Prints the mismatch count.
// Result on ROCm 7.2 / gfx950: 2 535 452 of 16 777 216 differ -- the single-precision root is a
// ~1 ulp v_sqrt_f32, not the IEEE root, so the kernels keep the double-precision form.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chk(unsigned long long *bad)
{
    unsigned const i = blockIdx.x * blockDim.x + threadIdx.x;          // 2^24 values
    unsigned long long b = 0;
    for (unsigned e = 0; e < 2; e++) {
        float const x = __uint_as_float(((127u + e) << 23) | (i & 0x7fffffu));
        float const a = __fsqrt_rn(x);
        float const r = (float) sqrt((double) x);
        b += (__float_as_uint(a) != __float_as_uint(r));
    }
    if (b) atomicAdd(bad, b);
}
int main()
{
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8);
    hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chk, dim3((1u << 23) / 256), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mismatches: %llu of %u\n", h, 2u << 23);
    return h != 0;
}
