/*
 * lh_wave.h -- wave64 cooperation primitives used by the kernels.
 *
 * One wavefront (64 lanes, gfx950) owns one stream-channel; the 576 spectral
 * lines / 288 line pairs of a granule are spread over its lanes, scalar
 * decisions are wave-uniform, and the reductions the reference does with
 * serial loops (ix_max, bit sums, "any distorted band") become cross-lane
 * reductions.  All primitives here must be called from wave-uniform control
 * flow.
 *
 * LH_EMU selects the CPU fiber emulation used by tests/hipemu (test tool only).
 */
#ifndef LH_WAVE_H
#define LH_WAVE_H

#include <stdint.h>

#ifdef LH_EMU
/* ------------------------------------------------------------------ */
#include "hipemu.h"

static inline int lh_lane(void) { return hipemu_lane(); }
static inline int lh_wave_id(void) { return hipemu_wave(); }
#define LH_WAVE_SYNC() hipemu_wave_sync()

static inline uint32_t
lh_wave_sum_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0;
    for (int i = 0; i < 64; i++)
        s += (uint32_t) x[i];
    return s;
}

static inline uint64_t
lh_wave_sum_u64(uint64_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint64_t s = 0;
    for (int i = 0; i < 64; i++)
        s += x[i];
    return s;
}

static inline uint32_t
lh_wave_max_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0;
    for (int i = 0; i < 64; i++)
        if ((uint32_t) x[i] > s)
            s = (uint32_t) x[i];
    return s;
}

static inline uint32_t
lh_wave_min_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0xffffffffu;
    for (int i = 0; i < 64; i++)
        if ((uint32_t) x[i] < s)
            s = (uint32_t) x[i];
    return s;
}

static inline uint64_t
lh_wave_or_u64(uint64_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint64_t s = 0;
    for (int i = 0; i < 64; i++)
        s |= x[i];
    return s;
}

static inline uint64_t
lh_ballot(int pred)
{
    const uint64_t *x = hipemu_wave_exchange(pred ? 1 : 0);
    uint64_t s = 0;
    for (int i = 0; i < 64; i++)
        if (x[i])
            s |= (1ull << i);
    return s;
}

static inline uint32_t
lh_bcast_u32(uint32_t v, int src)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    return (uint32_t) x[src];
}

static inline int lh_ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
static inline int lh_popc64(uint64_t m) { return __builtin_popcountll(m); }
static inline double lh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

#else
/* ------------------------------------------------------------------ */
#include <hip/hip_runtime.h>

__device__ __forceinline__ int lh_lane(void) { return (int) (threadIdx.x & 63); }
__device__ __forceinline__ int lh_wave_id(void) { return (int) (threadIdx.x >> 6); }

/* LDS traffic of one wave is executed in order; this only stops the compiler
 * from moving LDS accesses across the point where lanes exchange data through
 * LDS, and waits for outstanding LDS operations. */
#define LH_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
                            __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ uint32_t
lh_wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += (uint32_t) __shfl_xor((int) v, o, 64);
    return v;
}

__device__ __forceinline__ uint64_t
lh_wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += (uint64_t) __shfl_xor((long long) v, o, 64);
    return v;
}

__device__ __forceinline__ uint32_t
lh_wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = (uint32_t) __shfl_xor((int) v, o, 64);
        v = (w > v) ? w : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t
lh_wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = (uint32_t) __shfl_xor((int) v, o, 64);
        v = (w < v) ? w : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t
lh_wave_or_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v |= (uint64_t) __shfl_xor((long long) v, o, 64);
    return v;
}

__device__ __forceinline__ uint64_t lh_ballot(int pred) { return __ballot(pred); }

__device__ __forceinline__ uint32_t
lh_bcast_u32(uint32_t v, int src)
{
    return (uint32_t) __shfl((int) v, src, 64);
}

__device__ __forceinline__ int lh_ffs64(uint64_t m) { return m ? (__ffsll((long long) m) - 1) : -1; }
__device__ __forceinline__ int lh_popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ double lh_fma(double a, double b, double c) { return __fma_rn(a, b, c); }

#endif

#endif
