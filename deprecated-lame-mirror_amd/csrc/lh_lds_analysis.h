/*
 * lh_lds_analysis.h -- the LDS image of lh_analysis.hip's workgroups (included by lh_dev_common.h in place of the encode
 * kernel's LhLds): the psy model's scratch and little else, 18.9 KB -- eight workgroups per CU.
 */
struct LhLds {
    LhSmallState ss;            /* (named by lh_compute_masking's recurrence half, which is not instantiated there) */
    LhCtxShared ctx;
    LhRgSlot rg[2];
    int     uselong[2];
    int     pad[2];
    union __attribute__((aligned(16))) {
        LhPsyLds psy;
    } u;
};
__shared__ LhLds lh_lds __attribute__((aligned(16)));
