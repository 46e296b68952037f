/*
 * lh_dev_vbrold.h -- the old VBR loop, lame_set_VBR(vbr_rh) (reference quantize.c:1340-1578: get_framebits,
 * VBR_old_prepare, bitpressure_strategy, VBR_old_iteration_loop; :159-223 psfb21_analogsilence).
 *
 * Per granule and channel the loop looks for the smallest budget at which the CBR search leaves no band with
 * audible noise (lq_vbrold_body in lh_dev_qloop.h: the granule stays in registers through the whole bisection);
 * the frame takes the smallest bitrate that holds what the searches used.  Should even the largest frame be too
 * small, the allowed noise is raised, the budgets shrink, and every granule is searched again STARTING FROM the
 * scalefactors the last pass left -- so what a pass leaves before the finishing steps (scalefactor storage,
 * Huffman region split) is parked per granule in LDS (LhVbrOldSave), while the finished granule goes out as usual;
 * a later pass simply overwrites it.  Wave = channel, as everywhere.
 */
#ifndef LH_DEV_VBROLD_H
#define LH_DEV_VBROLD_H
#ifndef LH_VBROLD_FRAMEFN
#define LH_VBROLD_FRAMEFN LH_DEVFN
#endif

/* one span of lines [lo, hi) above the last scalefactor band, made of the parts bound[0..n] (positions counted from
 * bound[0] at line lo) with one threshold each: from the top line downwards everything below its threshold becomes
 * zero, up to the first line that is not (reference quantize.c:168-186) */
LH_DEVFN void
lh_silence_span(const LhCtx & c, float *xr, int lo, int hi, const int *bound, const float thr[LH_PSFB21])
{
    for (int base = hi - 1; base >= lo; base -= 64) {
        int const j = base - c.lane;
        int const in = j >= lo;
        int const pos = (in ? j : lo) - lo + bound[0];
        float   t = thr[0];
        uint64_t loud;
#pragma unroll
        for (int k = 1; k < LH_PSFB21; k++)
            t = (pos >= bound[k]) ? thr[k] : t;
        loud = lh_ballot(in && !(lh_fabsf(xr[in ? j : lo]) < t));
        if (loud) {
            int const first = lh_ffs64(loud);   /* the lowest lane = the highest such line */
            if (c.lane < first)
                xr[j] = 0.0f;
            break;
        }
        if (in)
            xr[j] = 0.0f;
    }
    LH_WAVE_SYNC();
}

/* reference quantize.c:159-223; the spectrum is already in window-major order for short blocks */
LH_DEVFN void
lh_psfb21_silence(const LhCtx & c, const LhQR & R, float *xr)
{
    const LhTables *T = c.T;
    float const adj = lh_lds.ss.ath_adjust_factor;
    float   thr[LH_PSFB21];
    LH_WAVE_SYNC();
    if (R.block_type != LH_SHORT_TYPE) {
        float const fact = T->longfact[21];
#pragma unroll
        for (int k = 0; k < LH_PSFB21; k++) {
            float   a = lh_ath_adjust(T, adj, T->ath_psfb21[k], T->ath_floor, 0);
            if (fact > 1e-12f)
                a *= fact;
            thr[k] = a;
        }
        lh_silence_span(c, xr, T->psfb21[0], T->psfb21[LH_PSFB21], T->psfb21, thr);
    }
    else {
        float const fact = T->shortfact[12];
        int const wd = T->sfb_s[13] - T->sfb_s[12];
#pragma unroll
        for (int k = 0; k < LH_PSFB12; k++) {
            float   a = lh_ath_adjust(T, adj, T->ath_psfb12[k], T->ath_floor, 0);
            if (fact > 1e-12f)
                a *= fact;
            thr[k] = a;
        }
        for (int block = 0; block < 3; block++) {
            int const lo = T->sfb_s[12] * 3 + wd * block;
            lh_silence_span(c, xr, lo, lo + (T->psfb12[LH_PSFB12] - T->psfb12[0]), T->psfb12, thr);
        }
    }
}

/* ---- one granule of one channel (wave) ----
 * pass 0: VBR_old_prepare's per-granule part (geometry, analog silence at the top of the spectrum, allowed noise),
 * then the search and the finishing steps; pass > 0: the search once more from what the last pass parked, with the
 * raised noise allowance (LhVbrOldSave.xmin, scaled by the frame function). */
LH_DEVFN void
lh_vbrold_granule(int qch, int gr, int rch, int pass, int min_bits, int max_bits, int substep, LhGranule * o,
                  const int8_t * g0sf)
{
    LhCtx const c = lh_ctx_load();
    LhLds & L = lh_lds;
    LhChanLds & Q = L.u.quant.ch[qch];
    LhVbrOldSave & sv = L.old[gr][qch];
    float  *xr = L.xr[qch][gr];
    int const s = c.lane;
    int const band = s <= LH_SFBMAX ? s : LH_SFBMAX;
    LhQR    R;
    LhGrR   g;
    int     nonzero, live;
    gr = lh_uni_i(gr);
    pass = lh_uni_i(pass);
    min_bits = lh_uni_i(min_bits);
    max_bits = lh_uni_i(max_bits);

    /* The heavy steps are the out-of-line stages the CBR frame loop uses (R / g travel through the channel's LDS slot):
     * inlined here they made this function spill and save ~35 callee-saved registers per call through scratch memory,
     * which at four workgroups per CU falls out of the L2 -- 160 KB of HBM traffic per frame (profiles/r03_pmc_vbrold2.json). */
    lh_init_outer_loop(qch, gr, lh_uni_i(L.block_type[gr][qch]), lh_uni_i(substep), pass == 0);
    if (pass == 0) {
        R = lh_uniform(L.rg[qch].R);
        lh_psfb21_silence(c, R, xr);
        lh_calc_xmin(qch, gr, rch);
        if (s <= LH_SFBMAX)
            sv.xmin[s] = Q.l3_xmin[s];
        R = lh_uniform(L.rg[qch].R);
        g = lh_uniform(L.rg[qch].g);
    }
    else {
        /* the granule as the last pass left it */
        R = lh_uniform(sv.R);
        g = lh_uniform(sv.g);
        LH_WAVE_SYNC();
        if (s <= LH_SFBMAX) {
            Q.sf[0][s] = sv.sf[s];
            Q.l3_xmin[s] = sv.xmin[s];
        }
        LH_WAVE_SYNC();
    }
    nonzero = lh_init_xrpow(c, Q, R, g, xr);
    live = nonzero && max_bits != 0;
    if (live) {
        lh_zero_tail(c, Q, R);
        lh_rg_put(c, R, g);
        int const cls = lh_uni_i(lh_vbrold_class(c, R.block_type, R.substep_shaping));
        if (lq_needs_tail(c, Q, R)) {
            if (cls == 2)
                lq_vbrold_stage5n(qch, gr, min_bits, max_bits, pass != 0);
            else if (cls == 1)
                lq_vbrold_stage5m(qch, gr, min_bits, max_bits, pass != 0);
            else
                lq_vbrold_stage5(qch, gr, min_bits, max_bits, pass != 0);
        }
        else {
            if (cls == 2)
                lq_vbrold_stage4n(qch, gr, min_bits, max_bits, pass != 0);
            else if (cls == 1)
                lq_vbrold_stage4m(qch, gr, min_bits, max_bits, pass != 0);
            else
                lq_vbrold_stage4(qch, gr, min_bits, max_bits, pass != 0);
        }
        R = lh_uniform(L.rg[qch].R);
        g = lh_uniform(L.rg[qch].g);
    }
    else if (nonzero) {
        /* energy but no bits (cannot happen with the reference's on_pe): nothing is coded */
        for (int i = s; i < 576; i += 64)
            Q.ix[0][i] = 0;
        LH_WAVE_SYNC();
    }
    /* park what a further pass starts from */
    LH_WAVE_SYNC();
    if (s <= LH_SFBMAX)
        sv.sf[s] = (uint8_t) Q.sf[0][band];
    if (s == 0) {
        sv.R = R;
        sv.g = g;
        sv.used_bits = live ? g.part2_3_length + g.part2_length : 0;
        if (pass == 0)
            sv.ath_over = R.ath_over;
    }
    LH_WAVE_SYNC();
    /* iteration_finish_one (reference quantize.c:1213-1232) */
    lh_rg_put(c, R, g);
    lh_best_scalefac_store(qch, gr, g0sf, lh_uni_i(L.block_type[0][qch]));
    if (c.cfg->use_best_huffman == 1)
        lh_best_huffman_divide(qch);
    R = lh_uniform(L.rg[qch].R);
    g = lh_uniform(L.rg[qch].g);
    lh_store_granule(c, Q, R, g, xr, LH_AS_GLOBAL(LhGranule, o));
    if (lh_uni_i(lh_lds.ctx.bytes != nullptr))
        lh_emit_part_stage(qch, gr);        /* R / g are in the wave's LDS slot since the last stage call */
    if (s == 0)
        sv.fin_bits = g.part2_3_length + g.part2_length;
    LH_WAVE_SYNC();
}

/* ---- the frame (all waves).  In: pe_use through L.pe_use, ms_ener_ratio through L.ms_ener_ratio; out through
 * L.frame_bits (bitrate index), L.max_bits (bits used), L.mean_bits (ResvSize after the frame's bits),
 * L.targ_bits[0] (substep), L.pe_use[0][0] (what sv_qnt.masking_lower holds after the frame). ---- */
LH_VBROLD_FRAMEFN void
lh_vbrold_frame(LhFrameOut * fo_in, int mode_ext, int msoff)
{
    LhCtx const c = lh_ctx_load();
    LhFrameOut *fo = LH_AS_GLOBAL(LhFrameOut, fo_in);
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    int const w = c.wave, tid = c.tid;
    float   pe_use[2][2] = { {lh_uni_f(L.pe_use[0][0]), lh_uni_f(L.pe_use[0][1])},
    {lh_uni_f(L.pe_use[1][0]), lh_uni_f(L.pe_use[1][1])}
    };
    float const ms_ener_ratio[2] = { lh_uni_f(L.ms_ener_ratio[0]), lh_uni_f(L.ms_ener_ratio[1]) };
    int     ResvSize = lh_uni_i(lh_lds.ss.ResvSize), substep = lh_uni_i(lh_lds.ss.substep_shaping);
    int const maxi = cfg->vbr_max_bitrate_index, nch = cfg->channels;
    int     min_bits[2][2], max_bits[2][2];
    int     avg, top_bits, resv_top, dummy, bits = 0, bitrate_index = maxi, used_bits, analog_silence, fin;
    float   masking_lower;
    mode_ext = lh_uni_i(mode_ext);
    msoff = lh_uni_i(msoff);

    /* VBR_old_prepare: the budgets at the largest frame */
    top_bits = lh_vbr_full_bits(cfg, maxi, ResvSize, &dummy, &resv_top);
    avg = top_bits / LH_NGR;      /* (the reference divides ResvFrameBegin's return value, quantize.c:1409) */
    for (int gr = 0; gr < LH_NGR; gr++) {
        int const mxb = lh_on_pe(cfg, ResvSize, resv_top, &substep, pe_use[gr], max_bits[gr], avg, 0);
        if (mode_ext == LH_MPG_MD_MS_LR)
            lh_reduce_side(max_bits[gr], ms_ener_ratio[gr], avg, mxb);
        for (int ch = 0; ch < nch; ch++)
            bits += max_bits[gr][ch];
    }
    for (int gr = 0; gr < LH_NGR; gr++)
        for (int ch = 0; ch < 2; ch++) {
            if (ch >= nch)
                max_bits[gr][ch] = 0;
            if (bits > top_bits && bits > 0) {
                max_bits[gr][ch] *= top_bits;
                max_bits[gr][ch] /= bits;
            }
            min_bits[gr][ch] = 126 > max_bits[gr][ch] ? max_bits[gr][ch] : 126;
            max_bits[gr][ch] = lh_uni_i(max_bits[gr][ch]);
            min_bits[gr][ch] = lh_uni_i(min_bits[gr][ch]);
        }
    substep = lh_uni_i(substep);
    {
        /* sv_qnt.masking_lower as the frame's last granule / channel leaves it (quantize.c:1420-1428); only the
         * next frame's psycho-acoustic model reads it */
        int const ch = nch - 1, gl = LH_NGR - 1;
        float const pe = pe_use[gl][ch];
        if (L.block_type[gl][ch] != LH_SHORT_TYPE)
            masking_lower = lh_vbrold_masking_lower(cfg->mask_adjust - lh_vbrold_adjust(pe, 0));
        else
            masking_lower = lh_vbrold_masking_lower(cfg->mask_adjust_short - lh_vbrold_adjust(pe, 1));
    }
    LH_SYNC_WG();
    if (mode_ext == LH_MPG_MD_MS_LR) {
        float const k = (float) (LH_SQRT2 * 0.5);
        for (int i = tid; i < 2 * 576; i += LH_NT) {
            int const gr = i >= 576, j = i - 576 * gr;
            float const l = L.xr[0][gr][j];
            float const r = L.xr[1][gr][j];
            L.xr[0][gr][j] = (l + r) * k;
            L.xr[1][gr][j] = (l - r) * k;
        }
    }
    LH_SYNC_WG();
    for (int pass = 0;; pass++) {
        for (int gr = 0; gr < LH_NGR; gr++) {
            if (w < nch)
                lh_vbrold_granule(w, gr, msoff + w, pass, min_bits[gr][w], max_bits[gr][w], substep, &fo->gr[gr][w],
                                  fo->gr[0][w].scalefac);
            else if (pass == 0) {
                /* mono: no second channel, its payload slot is all zero */
                uint32_t *z = (uint32_t *) &fo->gr[gr][w];
                for (int i = c.lane; i < (int) (sizeof(LhGranule) / 4); i += 64)
                    z[i] = 0u;
            }
        }
        LH_SYNC_WG();
        used_bits = 0;
        fin = 0;
        analog_silence = 1;
        for (int gr = 0; gr < LH_NGR; gr++)
            for (int ch = 0; ch < nch; ch++) {
                LhVbrOldSave const &sv = L.old[gr][ch];
                used_bits += lh_uni_i(sv.used_bits);
                fin += lh_uni_i(sv.fin_bits);
                if (lh_uni_i(sv.ath_over))
                    analog_silence = 0;
            }
        /* the smallest frame that holds them */
        {
            int     i = (analog_silence && !cfg->enforce_min_bitrate) ? 1 : cfg->vbr_min_bitrate_index;
            for (; i < maxi; i++)
                if (used_bits <= lh_vbr_full_bits(cfg, i, ResvSize, &dummy, &dummy))
                    break;
            bitrate_index = lh_uni_i(i);
        }
        {
            int     fits = used_bits <= lh_vbr_full_bits(cfg, bitrate_index, ResvSize, &dummy, &dummy);
#ifdef LH_EMU
            /* test hook of the CPU emulator build only (see oracle/orc_vbr_old.c): real input does not get here */
            if (getenv("LH_TEST_FORCE_PRESSURE") && pass < atoi(getenv("LH_TEST_FORCE_PRESSURE")))
                fits = 0;
#endif
            if (fits)
                break;
        }
        if (pass >= 200) {
            /* (cannot happen: the budgets shrink by a tenth per pass) -- never hang the device */
            if (tid == 0)
                lh_lds.ss.status |= 16;
            break;
        }
        /* bitpressure_strategy (reference quantize.c:1456-1480): more noise allowed towards the top, smaller budgets */
        LH_SYNC_WG();
        for (int gr = 0; gr < LH_NGR; gr++) {
            if (w < nch) {
                LhVbrOldSave & sv = L.old[gr][w];
                int const short_block = lh_uni_i(sv.R.block_type) == LH_SHORT_TYPE;
                int const psy_lmax = lh_uni_i(sv.R.psy_lmax), psymax = lh_uni_i(sv.R.psymax), smin = lh_uni_i(sv.R.sfb_smin);
                int const s = c.lane;
                if (s < psy_lmax)
                    sv.xmin[s] *= 1. + .029 * s * s / LH_SBMAX_L / LH_SBMAX_L;
                else if (short_block && s < psy_lmax + 3 * (LH_SBMAX_S - smin) && s <= LH_SFBMAX) {
                    int const sfb = smin + (s - psy_lmax) / 3;
                    sv.xmin[s] *= 1. + .029 * sfb * sfb / LH_SBMAX_S / LH_SBMAX_S;
                }
                (void) psymax;
            }
            for (int ch = 0; ch < 2; ch++) {
                double const m = 0.9 * max_bits[gr][ch];
                max_bits[gr][ch] = lh_uni_i((int) ((min_bits[gr][ch] > m) ? (double) min_bits[gr][ch] : m));
            }
        }
        LH_SYNC_WG();
    }
    ResvSize -= fin;
    LH_SYNC_WG();
    if (tid == 0) {
        L.frame_bits = bitrate_index;
        L.max_bits = fin;
        L.mean_bits = ResvSize;
        L.targ_bits[0] = substep;
        L.pe_use[0][0] = masking_lower;
    }
    LH_SYNC_WG();
}

#endif
