#!/bin/bash
# Build A/B variants of the library: tools/build_variants.sh name1:"-DX=0 -DY=1" name2:"..." -> lamehip/liblamehip_<name>.so
# (same objects as the product library except lh_kernels.o; run tools/ab.sh on the GPU box to compare them)
set -e
cd "$(dirname "$0")/../deprecated-lame-mirror_amd/csrc"
make -s lh_kernels.o lh_kernels_vbr.o lh_kernels_lsf.o lh_api.o lh_host_init.o lh_bitstream.o lh_vbrtag.o lh_resample.o lh_replaygain.o
KOPT="-O2 -fno-slp-vectorize -falign-functions=256 ${KSCHED--mllvm -amdgpu-sched-strategy=iterative-ilp}"   # KSCHED= (empty) for the default strategy
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $KOPT -std=c++17 -fno-fast-math -ffp-contract=off -fPIC -I. -I../../include $defs -c lh_kernels.hip -o /tmp/lh_kernels_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lamehip/liblamehip_$name.so /tmp/lh_kernels_$name.o lh_kernels_vbr.o lh_kernels_lsf.o lh_api.o lh_host_init.o lh_bitstream.o lh_vbrtag.o lh_resample.o lh_replaygain.o -lm &&
    echo "built liblamehip_$name.so ($defs)" ) &
done
wait
