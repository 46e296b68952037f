#!/bin/bash
# development aid: one variant of the split pipeline's encode kernel object linked into a library of its own.
# usage: tools/mkvar.sh <name> [-DFLAG ...]   ->  deprecated-lame-mirror_amd/lamehip/liblamehip_<name>.so
# (VAR_SRC=analysis|subband builds a variant of that front kernel's MPEG-1 object instead)
set -e
N=$1; shift
cd "$(dirname "$0")/../deprecated-lame-mirror_amd/csrc"
HIPCC=/opt/rocm/bin/hipcc
COMMON="--offload-arch=gfx950 ${VAR_OPT:--O2} -fno-slp-vectorize -falign-functions=256 -std=c++17 -fno-fast-math -ffp-contract=off -fPIC -I. -I../../include"
Q=lh_kernels_q.o; A=lh_analysis.o; S=lh_subband.o
case "${VAR_SRC:-q}" in
  q) $HIPCC $COMMON ${VAR_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} -DLH_SPLIT "$@" -c lh_kernels.hip -o /tmp/var_$N.o; Q=/tmp/var_$N.o;;
  analysis) $HIPCC $COMMON "$@" -c lh_analysis.hip -o /tmp/var_$N.o; A=/tmp/var_$N.o;;
  subband) $HIPCC $COMMON "$@" -c lh_subband.hip -o /tmp/var_$N.o; S=/tmp/var_$N.o;;
esac
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../lamehip/liblamehip_$N.so lh_kernels.o lh_kernels_vbr.o lh_kernels_lsf.o $Q lh_kernels_q_vbr.o lh_kernels_q_lsf.o $A lh_analysis_lsf.o $S lh_subband_lsf.o lh_api.o lh_host_init.o lh_bitstream.o lh_vbrtag.o lh_resample.o lh_replaygain.o -lm
echo built liblamehip_$N.so
