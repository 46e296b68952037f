"""ctypes mirrors of include/lamehip_types.h (keep in sync; sizes are asserted
against the C side by tests/test_abi.py)."""
import ctypes as C

SBMAX_L, SBMAX_S, PSFB21, PSFB12, SFBMAX, CBANDS = 22, 13, 6, 6, 39, 64
BLKSIZE, BLKSIZE_S, PRECALC, QMAX, QMAX2, S3_MAX = 1024, 256, 8208, 257, 116, 1280


class LhConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "version samplerate samplerate_index bitrate_index avg_bitrate mode mode_gr channels vbr "
        "quality noise_shaping noise_shaping_amp noise_shaping_stop subblock_gain use_best_huffman "
        "full_outer_loop substep_shaping quant_comp quant_comp_short sfb21_extra short_blocks "
        "use_safe_joint_stereo use_temporal_masking force_ms sideinfo_len buffer_constraint "
        "frac_SpF disable_reservoir error_protection copyright original extension emphasis "
        "lowpassfreq").split()] + [
        ("msfix", C.c_float), ("ATHfixpoint", C.c_float), ("ATH_offset_db", C.c_float),
        ("ATH_offset_factor", C.c_float), ("ATHcurve", C.c_float), ("ATHtype", C.c_int),
        ("minval", C.c_float), ("mask_adjust", C.c_float), ("mask_adjust_short", C.c_float),
        ("masking_lower_long", C.c_float), ("masking_lower_short", C.c_float),
        ("pcm_scale", C.c_float), ("interChRatio", C.c_float),
        ("vbr_q", C.c_int), ("vbr_min_bitrate_index", C.c_int), ("vbr_max_bitrate_index", C.c_int),
        ("enforce_min_bitrate", C.c_int), ("vbr_avg_bitrate_kbps", C.c_int), ("compression_ratio", C.c_float), ("pcm_mix", C.c_float),
        ("pcm_scale_r", C.c_float), ("highpassfreq", C.c_int), ("ath_flags", C.c_int)]


class LhPsyBand(C.Structure):
    _fields_ = [
        ("masking_lower", C.c_float * CBANDS), ("minval", C.c_float * CBANDS),
        ("rnumlines", C.c_float * CBANDS), ("mld_cb", C.c_float * CBANDS),
        ("mld", C.c_float * SBMAX_L), ("bo_weight", C.c_float * SBMAX_L),
        ("s3ind", (C.c_int * 2) * CBANDS), ("numlines", C.c_int * CBANDS),
        ("bm", C.c_int * SBMAX_L), ("bo", C.c_int * SBMAX_L),
        ("npart", C.c_int), ("n_sb", C.c_int), ("s3_count", C.c_int),
        ("s3_row", C.c_int * CBANDS), ("s3", C.c_float * S3_MAX)]


class LhTables(C.Structure):
    _fields_ = [
        ("sfb_l", C.c_int * (SBMAX_L + 1)), ("sfb_s", C.c_int * (SBMAX_S + 1)),
        ("psfb21", C.c_int * (PSFB21 + 1)), ("psfb12", C.c_int * (PSFB12 + 1)),
        ("line_pad0", C.c_int * 13), ("pow43", C.c_float * PRECALC), ("line_pad1", C.c_float * 16),
        ("vqthr", C.c_float * PRECALC), ("line_pad2", C.c_float * 16), ("vq3", (C.c_float * 4) * PRECALC),
        ("adj43asm", C.c_float * PRECALC),
        ("ipow20", C.c_float * QMAX), ("pow20", C.c_float * (QMAX + QMAX2 + 1)),
        ("bv_scf", C.c_int * 576),
        ("ath_l", C.c_float * SBMAX_L), ("ath_s", C.c_float * SBMAX_S),
        ("ath_psfb21", C.c_float * PSFB21), ("ath_psfb12", C.c_float * PSFB12),
        ("ath_cb_l", C.c_float * CBANDS), ("ath_cb_s", C.c_float * CBANDS),
        ("ath_eql_w", C.c_float * (BLKSIZE // 2)),
        ("ath_floor", C.c_float), ("ath_decay", C.c_float), ("aa_sensitivity_p", C.c_float),
        ("ath_use_adjust", C.c_int),
        ("longfact", C.c_float * SBMAX_L), ("shortfact", C.c_float * SBMAX_S),
        ("psy_l", LhPsyBand), ("psy_s", LhPsyBand), ("psy_l_to_s", LhPsyBand),
        ("attack_threshold", C.c_float * 4), ("decay", C.c_float),
        ("ma_max_i1", C.c_float), ("ma_max_i2", C.c_float),
        ("fft_window", C.c_float * BLKSIZE), ("fft_window_s", C.c_float * (BLKSIZE_S // 2)),
        ("fht_tw", ((C.c_float * 4) * 128) * 4),
        ("amp_filter", C.c_float * 32), ("log_table", C.c_float * 513),
        ("sfb_line_l", C.c_uint8 * 576), ("sfb_line_s", C.c_uint8 * 576), ("hgrid", C.c_uint32 * 704),
        ("qthr", C.c_float * 256), ("mask_mid", C.c_double * 10), ("bvpack", C.c_uint32 * 288)]


class LhGranule(C.Structure):
    _fields_ = [
        ("l3_enc", C.c_int16 * 576), ("scalefac", C.c_int8 * SFBMAX), ("pad0", C.c_int8),
        ("part2_3_length", C.c_int16), ("part2_length", C.c_int16), ("big_values", C.c_int16),
        ("count1", C.c_int16), ("global_gain", C.c_int16), ("scalefac_compress", C.c_int16),
        ("block_type", C.c_int8), ("mixed_block_flag", C.c_int8),
        ("table_select", C.c_int8 * 3), ("subblock_gain", C.c_int8 * 3),
        ("region0_count", C.c_int8), ("region1_count", C.c_int8), ("preflag", C.c_int8),
        ("scalefac_scale", C.c_int8), ("count1table_select", C.c_int8), ("sfbmax", C.c_int8),
        ("sfbdivide", C.c_int8), ("pad1", C.c_int8), ("count1bits", C.c_int16),
        ("pad2", C.c_int16)]


class LhFrameOut(C.Structure):
    _fields_ = [
        ("gr", (LhGranule * 2) * 2), ("scfsi", (C.c_int8 * 4) * 2),
        ("main_data_begin", C.c_int16), ("resvDrain_pre", C.c_int16),
        ("resvDrain_post", C.c_int16), ("bitrate_index", C.c_int8), ("padding", C.c_int8),
        ("mode_ext", C.c_int8), ("pad", C.c_int8 * 7), ("resv_size", C.c_int32),
        ("frame_bits", C.c_int32)]


class LhUserParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("samplerate channels brate mode quality vbr vbr_q samplerate_out abr_kbps "
                                       "force_ms disable_reservoir error_protection copyright original emphasis extension "
                                       "short_blocks strict_ISO lowpassfreq lowpasswidth").split()] + [
        ("scale", C.c_float), ("scale_left", C.c_float), ("scale_right", C.c_float)]


class LhInitAux(C.Structure):
    _fields_ = [(n, C.c_float) for n in "lowpass1 lowpass2 attackthre attackthre_s".split()] + [
        ("vbr_q", C.c_int), ("vbr_q_frac", C.c_float), ("athaa_sensitivity", C.c_float),
        ("adjust_sfb21_db", C.c_float)]


def struct_diff(a, b, prefix="", skip=()):
    """Return the list of leaf fields where two ctypes structures differ bitwise."""
    out = []
    for name, typ in a._fields_:
        if name in skip or name.startswith("pad"):
            continue
        va, vb = getattr(a, name), getattr(b, name)
        if isinstance(va, C.Structure):
            out += struct_diff(va, vb, prefix + name + ".", skip)
        elif isinstance(va, C.Array) and isinstance(_leaf(va), C.Structure):
            for idx, (ea, eb) in enumerate(zip(_flat(va), _flat(vb))):
                out += struct_diff(ea, eb, "%s%s[%d]." % (prefix, name, idx), skip)
        elif isinstance(va, C.Array):
            ba, bb = bytes(va), bytes(vb)
            if ba != bb:
                esz = C.sizeof(va) // max(1, _leaf_count(va))
                idx = [i // esz for i in range(0, len(ba), esz) if ba[i:i + esz] != bb[i:i + esz]]
                out.append((prefix + name, len(idx), idx[:8]))
        else:
            if bytes(C.c_double(va)) != bytes(C.c_double(vb)) and va != vb:
                out.append((prefix + name, va, vb))
    return out


def _leaf_count(arr):
    n = 1
    t = arr
    while isinstance(t, C.Array):
        n *= len(t)
        t = t[0]
    return n


def _leaf(arr):
    t = arr
    while isinstance(t, C.Array):
        t = t[0]
    return t


def _flat(arr):
    if isinstance(arr, C.Array) and isinstance(arr[0], C.Array):
        for sub in arr:
            for e in _flat(sub):
                yield e
    else:
        for e in arr:
            yield e
