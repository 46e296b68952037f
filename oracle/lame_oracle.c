/*
 * lame_oracle.c -- single translation unit of the CPU oracle (TEST
 * INFRASTRUCTURE ONLY; see orc_common.h for the usage rule and parity status).
 */
#include "orc_psy.c"
#include "orc_mdct.c"
#include "orc_quant.c"
#include "orc_vbr.c"
#include "orc_abr.c"
#include "orc_vbr_old.c"
#include "orc_frame.c"
#include "orc_resample.c"
