/*
 * lh_dev_math.h -- device math leaves whose bits must equal the host libm of
 * the reference build.
 *
 * lh_powf(): the reference calls powf() inside the per-frame path
 * (athAdjust, reference quantize_pvt.c:554-573; NS_INTERP, psymodel.c:443-454).
 * On the x86-64 hosts this project targets, glibc 2.35 resolves powf to its
 * FMA variant of the "optimized routines" algorithm (log2 via a 16-entry table
 * + degree-5 polynomial, exp2 via a 32-entry table + cubic, all in double,
 * fused multiply-adds where the source has a*b+c).  That algorithm is restated
 * here with the same tables and the same fused operations so the device result
 * is bit-identical; tests/test_powf.py sweeps it against the host powf.
 * Domain: x >= 0 (both call sites guarantee it).
 *
 * lh_fast_log2(): table-driven log2 of the reference (util.c:976-1001); the
 * 513-entry table is built on the host (LhTables.log_table).
 */
#ifndef LH_DEV_MATH_H
#define LH_DEV_MATH_H

#include <stdint.h>
#include "lh_wave.h"

#ifndef LH_DEVFN
#ifdef LH_EMU
#define LH_DEVFN static inline
#define LH_STAGEFN static
#define LH_DEVCONST static const
#else
#define LH_DEVFN __device__ __forceinline__
/* stage-level functions are NOT inlined: one giant kernel body made the register
 * allocator spill thousands of dwords; per-stage allocation keeps the kernel at
 * two waves per SIMD */
#define LH_STAGEFN static __device__ __attribute__((noinline))
#define LH_DEVCONST __device__ static const
#endif
#endif

LH_DEVCONST uint64_t lh_powf_log2_tab[32] = {
    0x3ff661ec79f8f3beull, 0xbfdefec65b963019ull, 0x3ff571ed4aaf883dull, 0xbfdb0b6832d4fca4ull,
    0x3ff49539f0f010b0ull, 0xbfd7418b0a1fb77bull, 0x3ff3c995b0b80385ull, 0xbfd39de91a6dcf7bull,
    0x3ff30d190c8864a5ull, 0xbfd01d9bf3f2b631ull, 0x3ff25e227b0b8ea0ull, 0xbfc97c1d1b3b7af0ull,
    0x3ff1bb4a4a1a343full, 0xbfc2f9e393af3c9full, 0x3ff12358f08ae5baull, 0xbfb960cbbf788d5cull,
    0x3ff0953f419900a7ull, 0xbfaa6f9db6475fceull, 0x3ff0000000000000ull, 0x0000000000000000ull,
    0x3fee608cfd9a47acull, 0x3fb338ca9f24f53dull, 0x3feca4b31f026aa0ull, 0x3fc476a9543891baull,
    0x3feb2036576afce6ull, 0x3fce840b4ac4e4d2ull, 0x3fe9c2d163a1aa2dull, 0x3fd40645f0c6651cull,
    0x3fe886e6037841edull, 0x3fd88e9c2c1b9ff8ull, 0x3fe767dcf5534862ull, 0x3fdce0a44eb17bccull
};

LH_DEVCONST uint64_t lh_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull
};

LH_DEVFN double
lh_u64_as_f64(uint64_t u)
{
    union {
        uint64_t u;
        double  d;
    } c;
    c.u = u;
    return c.d;
}

LH_DEVFN uint64_t
lh_f64_as_u64(double d)
{
    union {
        uint64_t u;
        double  d;
    } c;
    c.d = d;
    return c.u;
}

LH_DEVFN uint32_t
lh_f32_as_u32(float f)
{
    union {
        uint32_t u;
        float   f;
    } c;
    c.f = f;
    return c.u;
}

LH_DEVFN float
lh_u32_as_f32(uint32_t u)
{
    union {
        uint32_t u;
        float   f;
    } c;
    c.u = u;
    return c.f;
}

/* powf for x >= 0, finite or +inf, y finite */
LH_DEVFN float
lh_powf(float x, float y)
{
    uint32_t ix = lh_f32_as_u32(x);
    uint32_t iy = lh_f32_as_u32(y);
    /* special cases that the main path cannot take */
    if ((iy << 1) == 0)
        return 1.0f;            /* x^0 */
    if (ix == 0x3f800000u)
        return 1.0f;            /* 1^y */
    if (ix == 0) {
        /* +0 ^ y */
        return (iy >> 31) ? lh_u32_as_f32(0x7f800000u) : 0.0f;
    }
    if (ix == 0x7f800000u)
        return (iy >> 31) ? 0.0f : lh_u32_as_f32(0x7f800000u);
    if (ix < 0x00800000u) {
        /* subnormal x: normalise */
        ix = lh_f32_as_u32(x * 8388608.0f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    {
        /* log2(x) in double */
        uint32_t tmp = ix - 0x3f330000u;
        int     i = (int) ((tmp >> 19) & 15u);
        uint32_t top = tmp & 0xff800000u;
        uint32_t iz = ix - top;
        int     k = ((int32_t) top) >> 23;
        double  invc = lh_u64_as_f64(lh_powf_log2_tab[2 * i]);
        double  logc = lh_u64_as_f64(lh_powf_log2_tab[2 * i + 1]);
        double  z = (double) lh_u32_as_f32(iz);
        double  r, y0, r2, yy, p, r4, q, logx, ylogx;
        double const A0 = lh_u64_as_f64(0x3fd27616c9496e0bull);   /*  0x1.27616c9496e0bp-2 */
        double const A1 = lh_u64_as_f64(0xbfd71969a075c67aull);   /* -0x1.71969a075c67ap-2 */
        double const A2 = lh_u64_as_f64(0x3fdec70a6ca7baddull);   /*  0x1.ec70a6ca7baddp-2 */
        double const A3 = lh_u64_as_f64(0xbfe7154748bef6c8ull);   /* -0x1.7154748bef6c8p-1 */
        double const A4 = lh_u64_as_f64(0x3ff71547652ab82bull);   /*  0x1.71547652ab82bp+0 */
        r = lh_fma(z, invc, -1.0);
        y0 = logc + (double) k;
        r2 = r * r;
        yy = lh_fma(A0, r, A1);
        p = lh_fma(A2, r, A3);
        r4 = r2 * r2;
        q = lh_fma(A4, r, y0);
        q = lh_fma(p, r2, q);
        yy = lh_fma(yy, r4, q);
        logx = yy;
        ylogx = (double) y *logx;
        if (((lh_f64_as_u64(ylogx) >> 47) & 0xffff) >= 0x80bf) {
            /* |y*log2(x)| >= 126 */
            if (ylogx > 127.99999995700433)
                return lh_u32_as_f32(0x7f800000u);      /* overflow */
            if (ylogx <= -150.0)
                return 0.0f;    /* underflow */
            if (ylogx < -149.0)
                return lh_u32_as_f32(0x00000001u);      /* may-underflow: 0x1.4p-75f squared, rounded */
        }
        {
            /* exp2 in double */
            double const SHIFT = lh_u64_as_f64(0x42e8000000000000ull);    /* 0x1.8p47 */
            double const C0 = lh_u64_as_f64(0x3fac6af84b912394ull);       /* 0x1.c6af84b912394p-5 */
            double const C1 = lh_u64_as_f64(0x3fcebfce50fac4f3ull);       /* 0x1.ebfce50fac4f3p-3 */
            double const C2 = lh_u64_as_f64(0x3fe62e42ff0c52d6ull);       /* 0x1.62e42ff0c52d6p-1 */
            double  kd = ylogx + SHIFT;
            uint64_t ki = lh_f64_as_u64(kd);
            double  rr, s, zz, rr2, e;
            uint64_t t;
            kd -= SHIFT;
            rr = ylogx - kd;
            t = lh_exp2f_tab[ki & 31u];
            t += ki << (52 - 5);
            s = lh_u64_as_f64(t);
            zz = lh_fma(C0, rr, C1);
            rr2 = rr * rr;
            e = lh_fma(C2, rr, 1.0);
            e = lh_fma(zz, rr2, e);
            e = e * s;
            return (float) e;
        }
    }
}

/* reference util.c:976-1001 */
LH_DEVFN float
lh_fast_log2(const float *log_table, float x)
{
    float   log2val, partial;
    uint32_t const bits = lh_f32_as_u32(x);
    int     mantisse = (int) (bits & 0x7fffffu);
    log2val = (float) ((int) ((bits >> 23) & 0xFFu) - 0x7f);
    partial = (float) (mantisse & ((1 << 14) - 1));
    partial *= 1.0f / ((1 << 14));
    mantisse >>= 14;
    log2val += log_table[mantisse] * (1.0f - partial) + log_table[mantisse + 1] * partial;
    return log2val;
}

/* the same with the table read through `tab(i)' (an LDS copy: LH_LOGT_LDS in lh_dev_common.h) */
#define LH_FAST_LOG2_VIA(tab, x, out) do { \
        uint32_t const bits_ = lh_f32_as_u32(x); \
        int     man_ = (int) (bits_ & 0x7fffffu); \
        float   l2_ = (float) ((int) ((bits_ >> 23) & 0xFFu) - 0x7f); \
        float   part_ = (float) (man_ & ((1 << 14) - 1)); \
        part_ *= 1.0f / ((1 << 14)); \
        man_ >>= 14; \
        l2_ += tab(man_) * (1.0f - part_) + tab(man_ + 1) * part_; \
        (out) = l2_; \
    } while (0)

#define LH_LOG2_OVER_LOG10 (0.69314718055994530942 / 2.30258509299404568402)

/* ---- logf / log10f as the host libm computes them (glibc 2.35: logf = the table-driven double
 * evaluation of sysdeps/ieee754/flt-32/e_logf.c, FMA variant; log10f = the fdlibm formula
 * around it, e_log10f.c).  The VBR loop's quality-7 scalefactor guess truncates
 * c * log10f(xmin / bw) to an integer, so the last bit matters (reference vbrquantize.c:315-333).
 * tests/test_powf.py sweeps both against the host. */
LH_DEVCONST uint64_t lh_logf_tab[32] = {
    0x3ff661ec79f8f3beull, 0xbfd57bf7808caadeull,
    0x3ff571ed4aaf883dull, 0xbfd2bef0a7c06ddbull,
    0x3ff49539f0f010b0ull, 0xbfd01eae7f513a67ull,
    0x3ff3c995b0b80385ull, 0xbfcb31d8a68224e9ull,
    0x3ff30d190c8864a5ull, 0xbfc6574f0ac07758ull,
    0x3ff25e227b0b8ea0ull, 0xbfc1aa2bc79c8100ull,
    0x3ff1bb4a4a1a343full, 0xbfba4e76ce8c0e5eull,
    0x3ff12358f08ae5baull, 0xbfb1973c5a611cccull,
    0x3ff0953f419900a7ull, 0xbfa252f438e10c1eull,
    0x3ff0000000000000ull, 0x0000000000000000ull,
    0x3fee608cfd9a47acull, 0x3faaa5aa5df25984ull,
    0x3feca4b31f026aa0ull, 0x3fbc5e53aa362eb4ull,
    0x3feb2036576afce6ull, 0x3fc526e57720db08ull,
    0x3fe9c2d163a1aa2dull, 0x3fcbc2860d224770ull,
    0x3fe886e6037841edull, 0x3fd1058bc8a07ee1ull,
    0x3fe767dcf5534862ull, 0x3fd4043057b6ee09ull
};

LH_DEVFN float
lh_logf(float x)
{
    uint32_t ix = lh_f32_as_u32(x);
    if (ix == 0x3f800000u)
        return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u)
            return -__builtin_inff();
        if (ix == 0x7f800000u)
            return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u)
            return __builtin_nanf("");
        ix = lh_f32_as_u32(x * 0x1p23f);        /* subnormal: normalise */
        ix -= 23u << 23;
    }
    {
        uint32_t const tmp = ix - 0x3f330000u;
        int const i = (int) ((tmp >> 19) % 16u);
        int const k = (int32_t) tmp >> 23;
        uint32_t const iz = ix - (tmp & (0x1ffu << 23));
        double const z = (double) lh_u32_as_f32(iz);
        double const invc = lh_u64_as_f64(lh_logf_tab[2 * i]), logc = lh_u64_as_f64(lh_logf_tab[2 * i + 1]);
        double const A0 = lh_u64_as_f64(0xbfd00ea348b88334ull), A1 = lh_u64_as_f64(0x3fd5575b0be00b6aull), A2 = lh_u64_as_f64(0xbfdffffef20a4123ull);
        double const Ln2 = lh_u64_as_f64(0x3fe62e42fefa39efull);
        double const r = lh_fma(z, invc, -1.0);
        double const y0 = lh_fma((double) k, Ln2, logc);
        double const r2 = r * r;
        double  y = lh_fma(A1, r, A2);
        y = lh_fma(A0, r2, y);
        y = lh_fma(y, r2, y0 + r);
        return (float) y;
    }
}

LH_DEVFN float
lh_log10f(float x)
{
    float const two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f,
        log10_2lo = 7.9034151668e-07f;
    int32_t hx = (int32_t) lh_f32_as_u32(x), k = 0, i;
    float   y, z;
    if (hx < 0x00800000) {
        if ((hx & 0x7fffffff) == 0)
            return -__builtin_inff();
        if (hx < 0)
            return __builtin_nanf("");
        k -= 25;
        x *= two25;
        hx = (int32_t) lh_f32_as_u32(x);
    }
    if (hx >= 0x7f800000)
        return x + x;
    k += (hx >> 23) - 127;
    i = (int32_t) (((uint32_t) k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    y = (float) (k + i);
    x = lh_u32_as_f32((uint32_t) hx);
    z = y * log10_2lo + ivln10 * lh_logf(x);
    return z + y * log10_2hi;
}

/* ---- the two double-precision expressions of the old VBR loop (reference quantize.c:1420-1428) ----
 *   adjust        = (float) (1.28 / (1 + exp(3.5 - pe / 300.)) - 0.05)        (2.56 .. 0.14 for short blocks)
 *   masking_lower = (float) pow(10.0, masking_lower_db * 0.1)
 * Both take one float and give one float, so "equal to the host libm" can be checked for EVERY input: tools/
 * sweep_vbrold_math.c runs these functions against glibc's exp / pow over all floats of their domains (|pe| <
 * 2^20 -- beyond, exp is 0 or so large that the result no longer moves --, |db| <= 32) -- tests/test_vbrold_math.py runs a sample of that on every CPU test run.  exp here: the
 * usual reduction by ln 2 in two parts and a degree-14 Taylor polynomial, about 0.6 ulp in double, which is 2^29
 * times finer than the float the result is rounded to. */
LH_DEVFN double
lh_exp_dd(double hi, double lo)
{
    double const INVLN2 = lh_u64_as_f64(0x3ff71547652b82feull), LN2HI = lh_u64_as_f64(0x3fe62e42fee00000ull),
        LN2LO = lh_u64_as_f64(0x3dea39ef35793c76ull), SHIFT = lh_u64_as_f64(0x4338000000000000ull);
    double  kd = hi * INVLN2 + SHIFT, r, p;
    int const k = (int) (uint32_t) lh_f64_as_u64(kd);
    kd -= SHIFT;
    r = lh_fma(-kd, LN2HI, hi);
    r = lh_fma(-kd, LN2LO, r);
    r += lo;
    p = lh_u64_as_f64(0x3da93974a8c07c9dull);           /* 1 / 14! */
    p = p * r + lh_u64_as_f64(0x3de6124613a86d09ull);
    p = p * r + lh_u64_as_f64(0x3e21eed8eff8d898ull);
    p = p * r + lh_u64_as_f64(0x3e5ae64567f544e4ull);
    p = p * r + lh_u64_as_f64(0x3e927e4fb7789f5cull);
    p = p * r + lh_u64_as_f64(0x3ec71de3a556c734ull);
    p = p * r + lh_u64_as_f64(0x3efa01a01a01a01aull);
    p = p * r + lh_u64_as_f64(0x3f2a01a01a01a01aull);
    p = p * r + lh_u64_as_f64(0x3f56c16c16c16c17ull);
    p = p * r + lh_u64_as_f64(0x3f81111111111111ull);
    p = p * r + lh_u64_as_f64(0x3fa5555555555555ull);
    p = p * r + lh_u64_as_f64(0x3fc5555555555555ull);
    p = p * r + 0.5;                                    /* 1 / 2! */
    p = 1.0 + (r + (r * r) * p);
    {
        /* 2^k in two factors: the product overflows to infinity / underflows as IEEE arithmetic has it (pe may be
         * hugely negative -- the PE smoothing filter's scale can be -- and exp() of the reference is then +inf) */
        int const k1 = k / 2, k2 = k - k1;
        double const f1 = lh_u64_as_f64((uint64_t) (1023 + k1) << 52), f2 = lh_u64_as_f64((uint64_t) (1023 + k2) << 52);
        return (p * f1) * f2;
    }
}

LH_DEVFN float
lh_vbrold_adjust(float pe, int short_block)
{
    double const x = 3.5 - (double) pe / 300.;
    double const e = (x < -1000.) ? 0.0 : (x > 1000.) ? (double) __builtin_inff() : lh_exp_dd(x, 0.0);
    return short_block ? (float) (2.56 / (1 + e) - 0.14) : (float) (1.28 / (1 + e) - 0.05);
}

LH_DEVFN float
lh_vbrold_masking_lower(float db)
{
    double const LN10HI = lh_u64_as_f64(0x40026bb1bbb55516ull), LN10LO = lh_u64_as_f64(0xbcaf48ad494ea3e9ull);
    double const t = (double) db * 0.1;
    double const hi = t * LN10HI;
    double const lo = lh_fma(t, LN10HI, -hi) + t * LN10LO;
    return (float) lh_exp_dd(hi, lo);
}

#endif
