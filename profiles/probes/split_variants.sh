#!/bin/bash
# timing probes of corr_volume_split_stream: the library rebuilt with knock-out flags (see the source): run-time MV_SPLIT_DBG bits
# (libmacvo_hip_split_probe.so) and compile-time -DMV_SPLIT_KNOCK=<bits> variants (libmacvo_hip_split_k<bits>.so)
set -e
cd "$(dirname "$0")/../../mac-vo_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000"
OBJS=$(ls build/*.o | grep -v corr_volume_split)
hipcc $FL -DMV_SPLIT_PROBE -c corr_volume_split.hip -o /tmp/cvs_probe.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/probes/libmacvo_hip_split_probe.so $OBJS /tmp/cvs_probe.o
for k in ${KNOCKS:-1 2 4 8 15}; do
  hipcc $FL -DMV_SPLIT_PROBE -DMV_SPLIT_KNOCK=$k -c corr_volume_split.hip -o /tmp/cvs_k$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/probes/libmacvo_hip_split_k$k.so $OBJS /tmp/cvs_k$k.o
done
