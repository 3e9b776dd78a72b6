#!/bin/bash
# Round-6 experiment build (this container, no GPU needed): libsdrk.so whose fused N = 65536 kernel is compiled with
# -DSDRK_FUSED_EXPERIMENT (the policy sweep of fft_fused64k.hip: workgroups per CU, cache-policy bits of the streamed accesses,
# the timing-only NOWAIT form — chosen per process from SDRK_FU_* environment variables).  Every other object is the product's.
#   experiments/fused64k_policy/build.sh            -> sdr-iq-visualizer_amd/lib_fuexp/libsdrk.so      (ring depth 1, the product's)
#   experiments/fused64k_policy/build.sh 2          -> sdr-iq-visualizer_amd/lib_fuexp_d2/libsdrk.so   (ring depth 2)
set -e
D=${1:-1}
cd "$(dirname "$0")/../../sdr-iq-visualizer_amd/csrc"
make -s -j8
NAME=fuexp${FU_NAME_SUFFIX:-}; [ "$D" != 1 ] && NAME=${NAME}_d$D
mkdir -p ../build_$NAME ../lib_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=on -Wall -Wno-unused-function \
    -DSDRK_FUSED_EXPERIMENT -DFU_RING_SLOTS_N=$D ${FU_EXTRA_FLAGS:-} -c fft_fused64k.hip -o ../build_$NAME/fft_fused64k.o
OBJS=$(ls ../build/*.o | grep -v fft_fused64k.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib_$NAME/libsdrk.so $OBJS ../build_$NAME/fft_fused64k.o
ls -la ../lib_$NAME/libsdrk.so
