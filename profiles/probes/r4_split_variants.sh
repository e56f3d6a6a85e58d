#!/bin/bash
# Round-4 A/B builds of corr_volume_split_stream (run in the build container; the .so files travel to the GPU box with the snapshot):
#   r3    the round-3 kernel (git 8adf055) against this round's other objects
#   base  this round's loader walk (one pointer move per item, no per-half counters)
#   imm   + LDS-DMA pieces in groups of four behind one M0 / scalar base (immediate offsets)
#   vacc  + f16x2 accumulators in VGPRs (asm MFMAs, true ping-pong, no v_accvgpr_read)
#   both  imm + vacc
set -e
cd "$(dirname "$0")/../../mac-vo_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000"
OBJS=$(ls build/*.o | grep -v corr_volume_split)
OUT=../../profiles/probes
build() { # name, source, flags...
  n=$1; src=$2; shift 2
  hipcc $FL "$@" -c $src -o /tmp/cvs_$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmacvo_hip_r4_$n.so $OBJS /tmp/cvs_$n.o
}
git show 8adf055:mac-vo_amd/csrc/corr_volume_split.hip > /tmp/corr_volume_split_r3.hip
build r3 /tmp/corr_volume_split_r3.hip
build base corr_volume_split.hip -DMV_SPLIT_DMA_IMM=0 -DMV_SPLIT_VACC=0
build imm corr_volume_split.hip -DMV_SPLIT_DMA_IMM=1 -DMV_SPLIT_VACC=0
build vacc corr_volume_split.hip -DMV_SPLIT_DMA_IMM=0 -DMV_SPLIT_VACC=1
build both corr_volume_split.hip -DMV_SPLIT_DMA_IMM=1 -DMV_SPLIT_VACC=1
ls -la $OUT/libmacvo_hip_r4_*.so
