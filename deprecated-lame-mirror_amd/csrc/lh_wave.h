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

/* a wave-uniform condition that is rarely true: the block behind it moves out of line, so that the common
 * path falls through its branch (a taken scalar branch costs ~28 cycles, one that falls through ~14:
 * tools/ubench/branch_cost.hip) */
#define LH_RARE(x) __builtin_expect(!!(x), 0)
#define LH_OFTEN(x) __builtin_expect(!!(x), 1)

#ifdef LH_EMU
/* ------------------------------------------------------------------ */
#include "hipemu.h"

static inline int lh_lane(void) { return hipemu_lane(); }
static inline int lh_wave_id(void) { return hipemu_wave(); }
#define LH_WAVE_SYNC() hipemu_wave_sync()
#define LH_WAVE_SYNC_MEM() hipemu_wave_sync()
#define LH_WAVE_ORDER() hipemu_wave_sync()
#define LH_SCHED_FENCE() do { } while (0)

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

static inline uint32_t
lh_wave_or_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0;
    for (int i = 0; i < 64; i++)
        s |= (uint32_t) x[i];
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

typedef struct { float x, y; } lh_f32x2;
typedef struct { float x, y, z, w; } lh_f32x4;

static inline float
lh_wave_max_f32(float v)
{
    union { float f; uint32_t u; } c;
    const uint64_t *x;
    float   m;
    c.f = v;
    x = hipemu_wave_exchange(c.u);
    c.u = (uint32_t) x[0];
    m = c.f;
    for (int i = 1; i < 64; i++) {
        c.u = (uint32_t) x[i];
        if (c.f > m)
            m = c.f;
    }
    return m;
}


/* several independent reductions at once (the device version interleaves their steps) */
template < int N > static inline void
lh_wave_sum_n(uint32_t (&v)[N])
{
    for (int i = 0; i < N; i++)
        v[i] = lh_wave_sum_u32(v[i]);
}

template < int N > static inline void
lh_wave_max_n(uint32_t (&v)[N])
{
    for (int i = 0; i < N; i++)
        v[i] = lh_wave_max_u32(v[i]);
}

/* packed 16-bit minimum of both halves */
static inline uint32_t
lh_pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t const al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16;
    return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16);
}


/* value of lane `src' (per-lane choice) */
static inline float
lh_shfl_f32(float v, int src)
{
    union { float f; uint32_t u; } c;
    const uint64_t *x;
    c.f = v;
    x = hipemu_wave_exchange(c.u);
    c.u = (uint32_t) x[src & 63];
    return c.f;
}

static inline uint32_t
lh_shfl_u32(uint32_t v, int src)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    return (uint32_t) x[src & 63];
}

/* see the device version below: lane r < 3 gets region r's totals, everyone the quadruples' total */
static inline uint32_t
lh_wave_sum_regions(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t q, uint32_t *L, uint32_t *H)
{
    uint32_t const p[3] = { p0, p1, p2 };
    uint32_t lo[3], hi[3], qt;
    int const me = lh_lane();
    for (int r = 0; r < 3; r++) {
        uint32_t const a = lh_wave_sum_u32(p[r] & 0x3ffu), b = lh_wave_sum_u32((p[r] >> 10) & 0x3ffu);
        lo[r] = a | (b << 16);
        hi[r] = lh_wave_sum_u32(p[r] >> 20);
    }
    qt = lh_wave_sum_u32(q);
    *L = me < 3 ? lo[me] : 0u;
    *H = me < 3 ? hi[me] : 0u;
    return qt;
}

/* wave maxima of eight words at once: lane k returns the maximum of word k & 7 (see the device version) */
static inline uint32_t
lh_wave_max8(const uint32_t (&v)[8])
{
    uint32_t t[8];
    for (int k = 0; k < 8; k++)
        t[k] = lh_wave_max_u32(v[k]);
    return t[lh_lane() & 7];
}

/* value of lane - n (0 where there is none); n = 1, 2, 3 */
template < int N > static inline uint32_t
lh_lane_minus_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    int const me = lh_lane();
    return me >= N ? (uint32_t) x[me - N] : 0u;
}

/* value of lane - D of the same row of 16 lanes (0 where there is none): the device's row_shr */
template < int D > static inline uint32_t
lh_row_shr_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    int const me = lh_lane();
    return (me & 15) >= D ? (uint32_t) x[me - D] : 0u;
}

/* value of the lane below (0 for lane 0) */
static inline uint32_t
lh_lane_below_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    int const me = lh_lane();
    return me > 0 ? (uint32_t) x[me - 1] : 0u;
}

/* value of the lane above; lane 63 gets lane 0's value of `next' (the wave's next slot of a striped array) */
static inline uint32_t
lh_lane_above_u32(uint32_t v, uint32_t next)
{
    const uint64_t *x = hipemu_wave_exchange(((uint64_t) next << 32) | v);
    int const me = lh_lane();
    return me < 63 ? (uint32_t) x[me + 1] : (uint32_t) (x[0] >> 32);
}


/* an integer sum and a float maximum (no NaNs) at once */
static inline void
lh_wave_sum_maxf(uint32_t a, float m, int *sum, float *mx)
{
    *sum = (int) lh_wave_sum_u32(a);
    *mx = lh_wave_max_f32(m);
}


/* minimum over lanes 0..15 */
static inline uint32_t
lh_row0_min_u32(uint32_t v)
{
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0xffffffffu;
    for (int i = 0; i < 16; i++)
        if ((uint32_t) x[i] < s)
            s = (uint32_t) x[i];
    return s;
}


/* a wave sum in two halves: the first three butterfly steps leave every lane with the sum over its
 * group of eight lanes, the second half finishes the wave total */
template < int N > static inline void
lh_wave_sum_head3(uint32_t (&v)[N])
{
    for (int i = 0; i < N; i++) {
        const uint64_t *x = hipemu_wave_exchange(v[i]);
        uint32_t s = 0;
        int const g = hipemu_lane() & ~7;
        for (int k = 0; k < 8; k++)
            s += (uint32_t) x[g + k];
        v[i] = s;
    }
}

template < int N > static inline void
lh_wave_sum_tail3(uint32_t (&v)[N])
{
    for (int i = 0; i < N; i++) {
        const uint64_t *x = hipemu_wave_exchange(v[i]);
        uint32_t s = 0;
        for (int k = 0; k < 64; k += 8)
            s += (uint32_t) x[k];
        v[i] = s;
    }
}

static inline void lh_lds_add(int *p, int v) { *p += v; }      /* fibers interleave only at sync points */
static inline void lh_lds_max(int *p, int v) { if (v > *p) *p = v; }
static inline void lh_lds_addf(float *p, float v) { *p += v; }
static inline int lh_uni_i(int v) { return v; }
static inline uint32_t lh_vec_u32(uint32_t v) { return v; }
static inline float lh_uni_f(float v) { return v; }
static inline long long lh_uni_ll(long long v) { return v; }
static inline double lh_uni_f64(double v) { return v; }
static inline int lh_ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
static inline int lh_popc64(uint64_t m) { return __builtin_popcountll(m); }
static inline int lh_clz64(uint64_t m) { return m ? __builtin_clzll(m) : 64; }
static inline int lh_clz32(uint32_t m) { return m ? __builtin_clz(m) : 32; }
static inline double lh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

#else
/* ------------------------------------------------------------------ */
#include <hip/hip_runtime.h>

__device__ __forceinline__ int lh_lane(void) { return (int) (threadIdx.x & 63); }
__device__ __forceinline__ int lh_wave_id(void) { return (int) (threadIdx.x >> 6); }

/* LDS traffic of one wave is executed in order; this only stops the compiler
 * from moving LDS accesses across the point where lanes exchange data through
 * LDS, and waits for outstanding LDS operations.  The fence names the LDS address space
 * only: HBM loads in flight (table look-ups) are not drained by it. */
#define LH_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local"); \
                            __builtin_amdgcn_wave_barrier(); } while (0)
/* LDS exchange inside one wave without draining it: a wave's DS instructions are executed in the
 * order they were issued, so a read that follows another lane's write in program order sees it;
 * all that is needed is that the compiler keeps that order (no s_waitcnt here -- the reads can be
 * in flight together with whatever follows) */
#define LH_WAVE_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); \
                             asm volatile("" ::: "memory"); } while (0)
/* nothing is scheduled across this point: keeps a batch of independent table look-ups together (all issued
 * before the first use) where the scheduler, short of registers, would otherwise string them into
 * load / wait / use chains */
#define LH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
/* (the psycho-acoustic model's name for the same thing: its lanes once exchanged data through the stream
 * state in HBM and needed global accesses drained; that state lives in LDS now, and draining the
 * table loads in flight at every phase change was what the model waited for) */
#define LH_WAVE_SYNC_MEM() LH_WAVE_SYNC()

/* Wave reductions on the DPP cross-lane network (no LDS round trips): four
 * full-permutation steps (quad_perm [1,0,3,2], quad_perm [2,3,0,1],
 * row_half_mirror, row_mirror) leave every lane of a 16-lane row with the row
 * total; the four row totals are then read with v_readlane and combined on the
 * scalar unit.  ~12 instructions instead of six ds_bpermute round trips.
 * lamehip_selftest() checks these against a serial LDS evaluation on the device. */
/* `old' = the operation's identity with bound_ctrl lets the compiler fold the lane permute into
 * the arithmetic instruction (v_add_u32_dpp ...): 4 VALU per reduction step chain instead of 8 */
template < int CTRL, uint32_t IDENT > __device__ __forceinline__ uint32_t
lh_dpp(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) IDENT, (int) v, CTRL, 0xf, 0xf, IDENT == 0u);
}

/* ... then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 leave the wave total
 * in lane 63: seven instructions per reduction */
template < int CTRL, int ROWMASK, uint32_t IDENT > __device__ __forceinline__ uint32_t
lh_dpp_rows(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) IDENT, (int) v, CTRL, ROWMASK, 0xf, false);
}

#define LH_DPP_REDUCE(OP, IDENT) \
    { uint32_t t_; \
      t_ = lh_dpp < 0xB1, IDENT > (v); v = OP(v, t_); \
      t_ = lh_dpp < 0x4E, IDENT > (v); v = OP(v, t_); \
      t_ = lh_dpp < 0x141, IDENT > (v); v = OP(v, t_); \
      t_ = lh_dpp < 0x140, IDENT > (v); v = OP(v, t_); \
      t_ = lh_dpp_rows < 0x142, 0xa, IDENT > (v); v = OP(v, t_); \
      t_ = lh_dpp_rows < 0x143, 0xc, IDENT > (v); v = OP(v, t_); \
      return (uint32_t) __builtin_amdgcn_readlane((int) v, 63); }

#define LH_OP_ADD(a, b) ((a) + (b))
#define LH_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define LH_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define LH_OP_OR(a, b)  ((a) | (b))

__device__ __forceinline__ uint32_t lh_wave_sum_u32(uint32_t v) LH_DPP_REDUCE(LH_OP_ADD, 0u)
__device__ __forceinline__ uint32_t lh_wave_max_u32(uint32_t v) LH_DPP_REDUCE(LH_OP_MAX, 0u)
__device__ __forceinline__ uint32_t lh_wave_min_u32(uint32_t v) LH_DPP_REDUCE(LH_OP_MIN, 0xffffffffu)
__device__ __forceinline__ uint32_t lh_wave_or_u32(uint32_t v) LH_DPP_REDUCE(LH_OP_OR, 0u)


/* several independent reductions at once: the steps of all chains are issued side by side, so a
 * chain's DPP latency is covered by the other chains instead of s_nops */
#define LH_DPP_STEP_N(OP, IDENT, CTRL) \
    _Pragma("unroll") for (int i_ = 0; i_ < N; i_++) { uint32_t const t_ = lh_dpp < CTRL, IDENT > (v[i_]); v[i_] = OP(v[i_], t_); }
#define LH_DPP_ROWS_N(OP, IDENT, CTRL, MASK) \
    _Pragma("unroll") for (int i_ = 0; i_ < N; i_++) { uint32_t const t_ = lh_dpp_rows < CTRL, MASK, IDENT > (v[i_]); v[i_] = OP(v[i_], t_); }

template < int N > __device__ __forceinline__ void
lh_wave_sum_n(uint32_t (&v)[N])
{
    LH_DPP_STEP_N(LH_OP_ADD, 0u, 0xB1) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x4E)
    LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x141) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x140)
    LH_DPP_ROWS_N(LH_OP_ADD, 0u, 0x142, 0xa) LH_DPP_ROWS_N(LH_OP_ADD, 0u, 0x143, 0xc)
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = (uint32_t) __builtin_amdgcn_readlane((int) v[i], 63);
}

template < int N > __device__ __forceinline__ void
lh_wave_max_n(uint32_t (&v)[N])
{
    LH_DPP_STEP_N(LH_OP_MAX, 0u, 0xB1) LH_DPP_STEP_N(LH_OP_MAX, 0u, 0x4E)
    LH_DPP_STEP_N(LH_OP_MAX, 0u, 0x141) LH_DPP_STEP_N(LH_OP_MAX, 0u, 0x140)
    LH_DPP_ROWS_N(LH_OP_MAX, 0u, 0x142, 0xa) LH_DPP_ROWS_N(LH_OP_MAX, 0u, 0x143, 0xc)
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = (uint32_t) __builtin_amdgcn_readlane((int) v[i], 63);
}


/* a wave sum in two halves (see lh_wave_sum_n): after the first three steps every lane holds the sum
 * over its group of eight lanes -- packed fields can be widened between the halves */
template < int N > __device__ __forceinline__ void
lh_wave_sum_head3(uint32_t (&v)[N])
{
    LH_DPP_STEP_N(LH_OP_ADD, 0u, 0xB1) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x4E) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x141)
}

template < int N > __device__ __forceinline__ void
lh_wave_sum_tail3(uint32_t (&v)[N])
{
    LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x140)
    LH_DPP_ROWS_N(LH_OP_ADD, 0u, 0x142, 0xa) LH_DPP_ROWS_N(LH_OP_ADD, 0u, 0x143, 0xc)
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = (uint32_t) __builtin_amdgcn_readlane((int) v[i], 63);
}

__device__ __forceinline__ uint32_t
lh_pk_min_u16(uint32_t a, uint32_t b)
{
    typedef unsigned short lh_u16x2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; lh_u16x2 v; } x, y, r;
    x.u = a;
    y.u = b;
    r.v = __builtin_elementwise_min(x.v, y.v);
    return r.u;
}


/* value of lane `src' (per-lane choice): ds_bpermute_b32 */
__device__ __forceinline__ float
lh_shfl_f32(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

__device__ __forceinline__ uint32_t
lh_shfl_u32(uint32_t v, int src)
{
    return (uint32_t) __builtin_amdgcn_ds_bpermute(src << 2, (int) v);
}

/* Wave totals of four packed words at once, delivered where count_bits needs them.  p0..p2 hold three
 * 10-bit fields a | b << 10 | c << 20 each (per-lane values below 2^7: eight lanes add up without a carry),
 * q two 16-bit fields.  Three butterfly steps sum every word over the lane's group of eight; then the
 * eight lanes of a group split the seven values that remain to be added across the groups among themselves
 * (lane k of a group: fields a | b << 16 of word k & 3 for k < 4, field c of word k & 3 for k >= 4; word 3 is
 * q) and three more steps on that ONE register finish all of them -- instead of three steps and a
 * v_readlane on each of seven registers.  Lane r < 3 returns word r's a | b << 16 in *L and its c in *H;
 * the total of q comes back in every lane. */
__device__ __forceinline__ uint32_t
lh_wave_sum_regions(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t q, uint32_t *L, uint32_t *H)
{
    enum { N = 4 };
    uint32_t v[N] = { p0, p1, p2, q };
    int const lane = lh_lane();
    LH_DPP_STEP_N(LH_OP_ADD, 0u, 0xB1) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x4E) LH_DPP_STEP_N(LH_OP_ADD, 0u, 0x141)
    uint32_t const w01 = (lane & 1) ? v[1] : v[0], w23 = (lane & 1) ? v[3] : v[2];
    uint32_t const w = (lane & 2) ? w23 : w01;          /* word (lane & 3) */
    int const is_q = (lane & 3) == 3;
    uint32_t const ab = is_q ? w : ((w & 0x3ffu) | ((w << 6) & 0x03ff0000u));
    uint32_t const cc = is_q ? 0u : (w >> 20);
    uint32_t x = (lane & 4) ? cc : ab;
    x += lh_dpp < 0x128, 0u > (x);      /* row_ror:8: the other group of eight of the row, same position */
    /* lane ^ 16 and lane ^ 32 on the vector unit (gfx950's row / half swaps of two registers: with the value in both, one
     * comes back as "mine or my partner's lower copy", the other as the upper copy, and their sum is the pair's total in
     * every lane) -- two instructions and a copy each instead of a round trip through the LDS crossbar on count_bits' chain */
    {
        auto const r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r16[0] + r16[1];
        auto const r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        x = r32[0] + r32[1];
    }
    *L = x;
    *H = lh_dpp < 0x104, 0u > (x);      /* row_shl:4: lane r reads lane r + 4 */
    return (uint32_t) __builtin_amdgcn_readlane((int) x, 3);
}

/* Wave maxima of eight words at once (the transposed scheme of lh_wave_sum_regions): three butterfly steps
 * per word inside the groups of eight lanes, then lane k of a group carries word k & 7 through the three
 * steps across the groups.  Lane k returns the maximum of word k & 7. */
__device__ __forceinline__ uint32_t
lh_wave_max8(const uint32_t (&w)[8])
{
    enum { N = 8 };
    uint32_t v[N];
    int const lane = lh_lane();
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = w[i];
    LH_DPP_STEP_N(LH_OP_MAX, 0u, 0xB1) LH_DPP_STEP_N(LH_OP_MAX, 0u, 0x4E) LH_DPP_STEP_N(LH_OP_MAX, 0u, 0x141)
    uint32_t const a = (lane & 1) ? v[1] : v[0], b = (lane & 1) ? v[3] : v[2];
    uint32_t const cc = (lane & 1) ? v[5] : v[4], d = (lane & 1) ? v[7] : v[6];
    uint32_t const ab = (lane & 2) ? b : a, cd = (lane & 2) ? d : cc;
    uint32_t x = (lane & 4) ? cd : ab;
    uint32_t t;
    t = lh_dpp < 0x128, 0u > (x);
    x = t > x ? t : x;
    {
        auto const r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r16[0] > r16[1] ? r16[0] : r16[1];
        auto const r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        x = r32[0] > r32[1] ? r32[0] : r32[1];
    }
    return x;
}

/* value of lane - n inside a row of 16 (0 where there is none); n = 1, 2, 3: row_shr */
template < int N > __device__ __forceinline__ uint32_t
lh_lane_minus_u32(uint32_t v)
{
    return lh_dpp < 0x110 + N, 0u > (v);
}

template < int D > __device__ __forceinline__ uint32_t
lh_row_shr_u32(uint32_t v)
{
    return lh_dpp < 0x110 + D, 0u > (v);
}

/* value of the lane below (0 for lane 0; within a row of 16, which is all count_bits asks for) */
__device__ __forceinline__ uint32_t
lh_lane_below_u32(uint32_t v)
{
    return lh_dpp < 0x111, 0u > (v);    /* row_shr:1 */
}

/* value of the lane above, across the whole wave (wave_shl:1); lane 63 gets lane 0's value of `next' -- the
 * wave's next slot of an array striped over the lanes (element lane + 64 k) */
__device__ __forceinline__ uint32_t
lh_lane_above_u32(uint32_t v, uint32_t next)
{
    /* lane 63 has no lane above: without bound_ctrl the shift leaves it what the destination held before -- `next's
     * lane 0, put there first */
    int const n0 = __builtin_amdgcn_readlane((int) next, 0);
    return (uint32_t) __builtin_amdgcn_update_dpp(n0, (int) v, 0x130, 0xf, 0xf, false);
}

/* an integer sum and a float maximum (no NaNs) at once: the two chains' steps side by side */
__device__ __forceinline__ void
lh_wave_sum_maxf(uint32_t a, float m, int *sum, float *mx)
{
    /* order-preserving map of the float to an unsigned integer */
    uint32_t const b = (uint32_t) __float_as_int(m);
    uint32_t k = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
#define LH_SM_STEP(CTRL) { uint32_t const ta_ = lh_dpp < CTRL, 0u > (a), tk_ = lh_dpp < CTRL, 0u > (k); a += ta_; k = tk_ > k ? tk_ : k; }
#define LH_SM_ROWS(CTRL, MASK) { uint32_t const ta_ = lh_dpp_rows < CTRL, MASK, 0u > (a), tk_ = lh_dpp_rows < CTRL, MASK, 0u > (k); a += ta_; k = tk_ > k ? tk_ : k; }
    LH_SM_STEP(0xB1) LH_SM_STEP(0x4E) LH_SM_STEP(0x141) LH_SM_STEP(0x140) LH_SM_ROWS(0x142, 0xa) LH_SM_ROWS(0x143, 0xc)
#undef LH_SM_STEP
#undef LH_SM_ROWS
    *sum = __builtin_amdgcn_readlane((int) a, 63);
    k = (uint32_t) __builtin_amdgcn_readlane((int) k, 63);
    *mx = __int_as_float((int) (k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)));
}


/* minimum over lanes 0..15: the four in-row steps only */
__device__ __forceinline__ uint32_t
lh_row0_min_u32(uint32_t v)
{
    uint32_t t_;
    t_ = lh_dpp < 0xB1, 0xffffffffu > (v); v = LH_OP_MIN(v, t_);
    t_ = lh_dpp < 0x4E, 0xffffffffu > (v); v = LH_OP_MIN(v, t_);
    t_ = lh_dpp < 0x141, 0xffffffffu > (v); v = LH_OP_MIN(v, t_);
    t_ = lh_dpp < 0x140, 0xffffffffu > (v); v = LH_OP_MIN(v, t_);
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 0);
}

typedef float2 lh_f32x2;
typedef float4 lh_f32x4;

/* maximum of 64 floats (no NaNs among them) */
__device__ __forceinline__ float
lh_wave_max_f32(float v)
{
#define LH_FMAX_STEP(CTRL) { float const t_ = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)); v = (t_ > v) ? t_ : v; }
    LH_FMAX_STEP(0xB1) LH_FMAX_STEP(0x4E) LH_FMAX_STEP(0x141) LH_FMAX_STEP(0x140)
#undef LH_FMAX_STEP
    {
        float const r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)),
            r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
            r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)),
            r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        float const a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
        return a > b ? a : b;
    }
}

__device__ __forceinline__ uint64_t
lh_wave_or_u64(uint64_t v)
{
    return (uint64_t) lh_wave_or_u32((uint32_t) v) | ((uint64_t) lh_wave_or_u32((uint32_t) (v >> 32)) << 32);
}

__device__ __forceinline__ uint64_t lh_ballot(int pred) { return __ballot(pred); }

/* value of lane `src' (src must be wave-uniform) */
__device__ __forceinline__ uint32_t
lh_bcast_u32(uint32_t v, int src)
{
    return (uint32_t) __builtin_amdgcn_readlane((int) v, src);
}

/* An object of the workgroup's LDS image by its LDS address (32 bits), and a word read through such an address:
 * address arithmetic on the integer keeps the access a ds_read with no generic-pointer detour */
__device__ __forceinline__ uint32_t
lh_lds_off(const void *p)
{
    return (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) char *) p;
}

__device__ __forceinline__ uint32_t
lh_lds_read_u32(uint32_t off)
{
    return *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) off;
}

/* a.lo * b.lo + a.hi * b.hi + c on unsigned 16-bit halves (v_dot2_u32_u16) */
__device__ __forceinline__ uint32_t
lh_dot2_u16(uint32_t a, uint32_t b, uint32_t c)
{
    typedef unsigned short lh_u16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(lh_u16x2_, a), __builtin_bit_cast(lh_u16x2_, b), c, false);
}

/* LDS atomics without a return value (ds_add_u32 / ds_max_i32) */
__device__ __forceinline__ void lh_lds_add(int *p, int v) { (void) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lh_lds_max(int *p, int v) { (void) __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
/* ds_add_f32: order of the additions is not defined -- only for sums whose use tolerates that */
__device__ __forceinline__ void lh_lds_addf(float *p, float v) { (void) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/* Tell the compiler that a value is wave-uniform (it is by construction, but came through
 * per-lane memory, which the compiler must treat as divergent): the value moves to a scalar
 * register and everything derived from it -- branches, address arithmetic -- is scalar. */
__device__ __forceinline__ int lh_uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* The opposite: a wave-uniform value that is about to be selected per lane.  With the operands in
 * scalar registers the compiler turns `lane == 0 ? a : lane == 1 ? b : c' into EXEC-mask branches
 * (a dozen scalar instructions and two branches per select); from vector registers it is two
 * v_cndmask. */
__device__ __forceinline__ uint32_t lh_vec_u32(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float lh_uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ long long lh_uni_ll(long long v)
{
    unsigned const lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (unsigned long long) v);
    unsigned const hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ((unsigned long long) v >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ double lh_uni_f64(double v) { return __longlong_as_double(lh_uni_ll(__double_as_longlong(v))); }
__device__ __forceinline__ int lh_ffs64(uint64_t m) { return m ? (__ffsll((long long) m) - 1) : -1; }
__device__ __forceinline__ int lh_popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int lh_clz64(uint64_t m) { return __clzll((long long) m); }
__device__ __forceinline__ int lh_clz32(uint32_t m) { return __clz((int) m); }
__device__ __forceinline__ double lh_fma(double a, double b, double c) { return __fma_rn(a, b, c); }

#endif

#endif
