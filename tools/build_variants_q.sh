#!/bin/bash
# A/B variants of the split pipeline's encode kernel object (lh_kernels_q.o): tools/build_variants_q.sh name1:"flags" name2:"flags" ...
# -> lamehip/liblamehip_<name>.so (every other object is the product's); flags replace "$KOPT $KSCHED"; -D switches may follow.
set -e
cd "$(dirname "$0")/../deprecated-lame-mirror_amd/csrc"
make -s -j8 all
OBJS="lh_kernels.o lh_kernels_vbr.o lh_kernels_lsf.o lh_kernels_q_vbr.o lh_kernels_q_lsf.o lh_analysis.o lh_analysis_lsf.o lh_subband.o lh_subband_lsf.o lh_api.o lh_host_init.o lh_bitstream.o lh_vbrtag.o lh_resample.o lh_replaygain.o"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $flags -std=c++17 -fno-fast-math -ffp-contract=off -fPIC -I. -I../../include -DLH_SPLIT -c lh_kernels.hip -o /tmp/lh_kernels_q_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lamehip/liblamehip_$name.so /tmp/lh_kernels_q_$name.o $OBJS -lm &&
    echo "built liblamehip_$name.so ($flags)" ) &
done
wait
