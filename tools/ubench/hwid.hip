// Development aid: which SIMD does wave i of a workgroup land on?  Launches B workgroups of T threads with L bytes
// of LDS each (so that the occupancy is the encode kernel's), every wave records HW_ID; prints, per wave index,
// the histogram of SIMD ids, and how many distinct waves share a (XCC, SE, CU, SIMD) at the moment of the launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <tuple>
#include <vector>
__global__ void k(unsigned *out, unsigned *xcc, int spin)
{
    extern __shared__ char lds[];
    unsigned v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if ((threadIdx.x & 63) == 0) {
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = v;
        xcc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = x;
    }
    // stay resident for a while so that the whole grid is co-resident
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long) spin) { }
    if (lds[threadIdx.x] == 77) out[0] = 0;
}
int main(int argc, char **argv)
{
    int B = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 256, L = argc > 3 ? atoi(argv[3]) : 40720;
    int W = T / 64;
    unsigned *d, *dx, *h = (unsigned *) malloc(B * W * 4), *hx = (unsigned *) malloc(B * W * 4);
    hipMalloc(&d, B * W * 4); hipMalloc(&dx, B * W * 4);
    hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, L);
    hipLaunchKernelGGL(k, dim3(B), dim3(T), L, 0, d, dx, 2000000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, B * W * 4, hipMemcpyDeviceToHost); hipMemcpy(hx, dx, B * W * 4, hipMemcpyDeviceToHost);
    int hist[8][4] = {{0}};
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned, unsigned>, int> per;
    for (int b = 0; b < B; b++)
        for (int w = 0; w < W; w++) {
            unsigned v = h[b * W + w], x = hx[b * W + w] & 15u;
            unsigned simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7;
            hist[w][simd]++;
            per[std::make_tuple(x, se, sh, cu, simd)]++;
        }
    for (int w = 0; w < W; w++)
        printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int cnt[64] = {0};
    for (auto &p : per) cnt[p.second < 63 ? p.second : 63]++;
    printf("distinct SIMDs used: %zu; waves per SIMD histogram:", per.size());
    for (int i = 0; i < 64; i++) if (cnt[i]) printf(" %dx%d", cnt[i], i);
    printf("\n");
    {   // who shares a CU: workgroup ids, and per wave the SIMD and the wave slot
        std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::vector<std::tuple<int, int, unsigned, unsigned>>> cus;
        for (int b = 0; b < B; b++)
            for (int w = 0; w < W; w++) {
                unsigned v = h[b * W + w], x = hx[b * W + w] & 15u;
                cus[std::make_tuple(x, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15)].push_back(std::make_tuple(b, w, (v >> 4) & 3, v & 15));
            }
        int shown = 0;
        for (auto &c : cus) {
            if (shown++ >= 6) break;
            printf("xcc %u se %u sh %u cu %u:", std::get<0>(c.first), std::get<1>(c.first), std::get<2>(c.first), std::get<3>(c.first));
            for (auto &e : c.second) printf("  wg %d w%d simd %u slot %u", std::get<0>(e), std::get<1>(e), std::get<2>(e), std::get<3>(e));
            printf("\n");
        }
        printf("CUs used: %zu\n", cus.size());
    }
    for (int b = 0; b < 4; b++) { printf("wg %d:", b); for (int w = 0; w < W; w++) printf(" %08x/x%u", h[b * W + w], hx[b * W + w] & 15u); printf("\n"); }
    return 0;
}
