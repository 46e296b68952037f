"""Arithmetic identities the CBR search (lh_dev_qloop.h) relies on instead of restating the reference's
expression literally.  Each is checked here exhaustively over the domain the kernel uses it on.

1. First rounding of the quantiser (reference takehiro.c:227-253, TAKEHIRO_IEEE754_HACK): the
   reference forms (float) ((double) x + 8388608.0); the kernel forms x + 8388608.0f.  Equal for every
   float 0 <= x < 8206.0 (IXMAX_VAL + 14: larger values are rejected before quantising).
2. The step tables scale by exact powers of two: pow20[k + 4] == 2 pow20[k], ipow20[k + 16] ==
   ipow20[k] / 8, which is what lets the kernel rebuild any entry from a few mantissas with ldexp
   (the host refuses to initialise otherwise: power_tables_scale_exactly, lh_host_init.c).
3. Second rounding of the quantiser: the reference adds adj43asm[k] (k = the first rounding) in double and
   rounds to float again; the kernel forms k - (x < qthr[k]) for k < 256 (LhTables.qthr, lh_host_init.c).
   Equal for every float whose first rounding is below 256.
4. mask_add outside the band around the diagonal (reference psymodel.c:323-341) asks whether the float
   quotient larger / smaller is below ma_max_i2; the kernel asks whether larger < midpoint(c, pred c) x smaller
   in double (lh_mask_add_far, lh_dev_psy.h).  Checked on pairs within a few ulps of the boundary, over the
   whole exponent range including denormals, and on random pairs.
5. mask_add near the diagonal (psymodel.c:323-333) multiplies the sum by table2[i], i = (int) (fast_log2(ratio)
   * 16 log10 2), or by 1 from ma_max_i1 on; the kernel counts how many of nine exact products
   LhTables.mask_mid[j] x smaller lie below larger (lh_mask_add_near).  The cell is walked over every float
   ratio below ma_max_i1 (it never decreases and steps at the eight floats the host found), then the whole
   rule is compared on boundary and random pairs.
"""
import numpy as np

import helpers  # noqa: F401  (puts the package on sys.path)


def test_float_addition_equals_the_double_rounding_for_every_quantisable_float():
    magic = np.float32(8388608.0)
    top = int(np.float32(8206.0).view(np.uint32))
    chunk = 1 << 20
    bad = 0
    for lo in range(0, top + 1, chunk):
        x = np.arange(lo, min(lo + chunk, top + 1), dtype=np.uint32).view(np.float32)
        ref = (x.astype(np.float64) + 8388608.0).astype(np.float32)
        mine = x + magic
        bad += int(np.count_nonzero(ref.view(np.uint32) != mine.view(np.uint32)))
    assert bad == 0


def test_step_tables_scale_by_exact_powers_of_two():
    import lamehip
    enc = lamehip.Encoder(44100, 128, require_device=False)
    T = enc.tables()
    pow20 = np.ctypeslib.as_array(T.pow20).astype(np.float32)
    ipow20 = np.ctypeslib.as_array(T.ipow20).astype(np.float32)
    assert np.array_equal(pow20[4:], pow20[:-4] * np.float32(2.0))
    assert np.array_equal(ipow20[16:], ipow20[:-16] / np.float32(8.0))


def test_float_root_equals_the_rounded_double_root():
    """amp_scalefac_bands forms (float) sqrt((double) trigger) (reference quantize.c:744); the kernel takes the
    correctly rounded float root.  Two binades exhaustively (the mantissa pattern repeats every two), plus
    a random sample of the range the trigger lives in."""
    x = np.arange(0x3f800000, 0x3f800000 + (1 << 24), dtype=np.uint32).view(np.float32)     # [1, 4)
    assert np.array_equal(np.sqrt(x.astype(np.float64)).astype(np.float32), np.sqrt(x))
    rng = np.random.default_rng(5)
    y = (rng.random(4_000_000, dtype=np.float32) * np.float32(1e9) + np.float32(1.0)).astype(np.float32)
    assert np.array_equal(np.sqrt(y.astype(np.float64)).astype(np.float32), np.sqrt(y))


def test_threshold_comparison_equals_the_second_rounding_for_every_quantisable_float():
    """LhTables.qthr (CBR search, k < 256: the quantised value), LhTables.vqthr and LhTables.vq3 (VBR noise search, every k; vq3 = threshold and the class's two values of pow43, what the kernel reads) against the reference's expression, for every float up to IXMAX + 0.5."""
    import lamehip
    enc = lamehip.Encoder(44100, 128, require_device=False)
    T = enc.tables()
    adj = np.ctypeslib.as_array(T.adj43asm).astype(np.float32)
    thr = np.ctypeslib.as_array(T.qthr).astype(np.float32)
    vqthr = np.ctypeslib.as_array(T.vqthr).astype(np.float32)
    vq3 = np.ctypeslib.as_array(T.vq3).astype(np.float32).reshape(-1, 4)
    pow43 = np.ctypeslib.as_array(T.pow43).astype(np.float32)
    enc.close()
    assert np.all(adj[1:256] < 0)           # what confines the class of k < 256 to the values k - 1 and k
    magic = np.float32(8388608.0)
    top = int(np.float32(8206.5).view(np.uint32))       # IXMAX_VAL + 0.5 (ties to even: 8206)
    head = int(np.float32(255.5).view(np.uint32))       # 255.5 ties to 256: the last float of class 255 is below it
    chunk = 1 << 20
    bad = bad_v = 0
    for lo in range(0, top + 1, chunk):
        x = np.arange(lo, min(lo + chunk, top + 1), dtype=np.uint32).view(np.float32)
        k = ((x + magic).view(np.uint32) - np.uint32(0x4B000000)).astype(np.int32)
        ref = ((x.astype(np.float64) + 8388608.0) + adj[k].astype(np.float64)).astype(np.float32)
        ref = (ref.view(np.uint32) - np.uint32(0x4B000000)).astype(np.int32)
        e = vqthr[k]
        mine_v = k - (x < np.abs(e)).astype(np.int32) + np.signbit(e).astype(np.int32)
        bad_v += int(np.count_nonzero(ref != mine_v))
        # LhTables.vq3: the same decision with the class's two values of pow43 beside the threshold (what the kernel reads)
        c = vq3[k]
        bad_v += int(np.count_nonzero(np.where(x < c[:, 0], c[:, 1], c[:, 2]) != pow43[ref]))
        if lo < head:
            n = min(len(x), head - lo)
            assert k[:n].max() < 256
            mine = k[:n] - (x[:n] < thr[k[:n]]).astype(np.int32)
            bad += int(np.count_nonzero(ref[:n] != mine))
    assert bad == 0 and bad_v == 0


def test_far_masking_rule_without_the_quotient():
    import lamehip
    enc = lamehip.Encoder(44100, 128, require_device=False)
    cs = [np.float32(enc.tables().ma_max_i2)]
    mid_i2 = float(enc.tables().mask_mid[9])
    enc.close()
    rng = np.random.default_rng(11)
    cs += [np.float32(v) for v in (1.0000001, 1.5, 3.1622777, 31.622776, 1000.0)]
    cs += list(rng.uniform(1.0, 100.0, 8).astype(np.float32))
    with np.errstate(over="ignore", divide="ignore", invalid="ignore", under="ignore"):
        for c in cs:
            below = (np.array([c], np.float32).view(np.uint32) - np.uint32(1)).view(np.float32)[0]
            bound = 0.5 * (np.float64(c) + np.float64(below))
            if c == cs[0]:
                assert bound == mid_i2      # what the kernel reads (LhTables.mask_mid[9])
            # smaller: every exponent (denormals included) with random mantissas; larger: c x smaller +- 0..40 ulps
            expo = np.repeat(np.arange(0, 254, dtype=np.uint32), 4000)
            lo = ((expo << 23) | rng.integers(0, 1 << 23, expo.size, dtype=np.uint32)).view(np.float32)
            base = (c * lo).astype(np.float32)
            ok = np.isfinite(base)
            lo, base = lo[ok], base[ok]
            hi = (base.view(np.uint32).astype(np.int64) + rng.integers(-40, 41, base.size)).clip(0, 0x7f7fffff).astype(np.uint32).view(np.float32)
            lo2 = rng.random(2_000_000, dtype=np.float32) * np.float32(1e6)
            hi2 = lo2 * (rng.random(2_000_000, dtype=np.float32) * np.float32(2.0) * c)
            lo = np.concatenate([lo, lo2, np.zeros(4, np.float32)])
            hi = np.concatenate([hi, hi2, np.array([0, 1, 1e-40, 3e38], np.float32)])
            hi, lo = np.maximum(hi, lo), np.minimum(hi, lo)
            ref = np.where(lo > 0, (hi / lo).astype(np.float32) < c, False)
            mine = hi.astype(np.float64) < bound * lo.astype(np.float64)
            assert np.array_equal(ref, mine), c


def _mask_cell(log_table, ratio):
    """the reference's table cell for float ratios >= 1 (util.c:976-1001, psymodel.c:331)"""
    bits = ratio.view(np.uint32)
    mant = (bits & np.uint32(0x7fffff)).astype(np.int32)
    whole = (((bits >> np.uint32(23)) & np.uint32(0xff)).astype(np.int32) - 0x7f).astype(np.float32)
    along = (mant & 16383).astype(np.float32) * np.float32(1.0 / 16384)
    slot = mant >> 14
    lg = whole + (log_table[slot] * (np.float32(1.0) - along) + log_table[slot + 1] * along)
    return (lg.astype(np.float64) * (np.float64(0.69314718055994530942 / 2.30258509299404568402) * np.float64(16.0))).astype(np.int32)


def test_near_masking_rule_without_quotient_or_logarithm():
    import lamehip
    enc = lamehip.Encoder(44100, 128, require_device=False)
    T = enc.tables()
    log_table = np.ctypeslib.as_array(T.log_table).astype(np.float32)
    mid = np.ctypeslib.as_array(T.mask_mid).astype(np.float64)
    c1 = np.float32(T.ma_max_i1)
    enc.close()
    table2 = np.array([1.33352 ** 2, 1.35879 ** 2, 1.38454 ** 2, 1.39497 ** 2, 1.40548 ** 2, 1.3537 ** 2, 1.30382 ** 2,
                       1.22321 ** 2, 1.14758 ** 2, 1.0]).astype(np.float32)       # psymodel.c:297-302

    def boundary(v):
        below = (np.array([v], np.float32).view(np.uint32) - np.uint32(1)).view(np.float32)[0]
        return 0.5 * (np.float64(v) + np.float64(below))

    # every float ratio in [1, ma_max_i1): cells 0..8, never decreasing, stepping where the host says
    lo_b, hi_b = int(np.float32(1.0).view(np.uint32)), int(c1.view(np.uint32))
    ratio = np.arange(lo_b, hi_b, dtype=np.uint32).view(np.float32)
    cell = _mask_cell(log_table, ratio)
    assert cell[0] == 0 and cell[-1] == 8 and np.all(np.diff(cell) >= 0)
    steps = ratio[1:][np.diff(cell) != 0]
    assert len(steps) == 8
    assert np.array_equal(np.array([boundary(r) for r in steps]), mid[:8])
    assert mid[8] == boundary(c1)
    # the whole rule on pairs around every boundary and on random pairs
    rng = np.random.default_rng(12)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore", under="ignore"):
        los, his = [], []
        for r in list(steps) + [c1]:
            expo = np.repeat(np.arange(0, 250, dtype=np.uint32), 800)
            lo = ((expo << 23) | rng.integers(0, 1 << 23, expo.size, dtype=np.uint32)).view(np.float32)
            base = (np.float32(r) * lo).astype(np.float32)
            ok = np.isfinite(base)
            lo, base = lo[ok], base[ok]
            hi = (base.view(np.uint32).astype(np.int64) + rng.integers(-40, 41, base.size)).clip(0, 0x7f7fffff).astype(np.uint32).view(np.float32)
            los.append(lo)
            his.append(hi)
        lo2 = rng.random(3_000_000, dtype=np.float32) * np.float32(1e6)
        los += [lo2, np.zeros(4, np.float32)]
        his += [lo2 * (rng.random(3_000_000, dtype=np.float32) * np.float32(5.0)), np.array([0, 1, 1e-40, 3e38], np.float32)]
        a, b = np.concatenate(his), np.concatenate(los)
        hi, lo = np.maximum(a, b), np.minimum(a, b)
        total = a + b
        q = (hi / np.where(lo > 0, lo, np.float32(1.0))).astype(np.float32)
        safe = np.where((lo > 0) & (q < c1), q, np.float32(1.0))
        ref = np.where(lo > 0, np.where(q >= c1, total, total * table2[_mask_cell(log_table, safe)]), hi)
        count = np.zeros(hi.size, np.int64)
        for j in range(9):
            count += hi.astype(np.float64) > mid[j] * lo.astype(np.float64)
        mine = total * table2[count]
        assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32))
