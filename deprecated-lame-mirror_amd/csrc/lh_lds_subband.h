/*
 * lh_lds_subband.h -- the LDS image of lh_subband.hip's workgroups (included by lh_dev_common.h in place of the encode
 * kernel's LhLds): the frame window / the spectra, the sub-band samples of three granules (29.3 KB).
 */
struct LhLds {
    LhCtxShared ctx;
    LhRgSlot rg[2];
    int     block_type[2][2];   /* [gr][ch] */
    float   mwin[4 * 36];       /* the MDCT windows and rotation constants (lh_mdct_win) */
    float   enw[288];           /* the polyphase window's coefficients (lh_enwindow), read a row per lane by the tap sums */
    union __attribute__((aligned(16))) {
        float   mf[2][LH_MF_PADDED];  /* (swizzled: LH_MF_SWZ) */
        float   xr[2][2][576];
    };
    union __attribute__((aligned(16))) {
        LhMdctLds mdct;
    } u;
};
__shared__ LhLds lh_lds __attribute__((aligned(16)));
