#!/bin/bash
# Builds the two probe libraries of the PGO kernel next to this script (run in the build container after `make -C mac-vo_amd/csrc`; the .so files
# are git-ignored and travel to the GPU box with the snapshot):
#   libmacvo_hip_pgostamps.so    -DMV_PGO_STAMPS          s_memtime stamps read by pgo_stamps.py
#   libmacvo_hip_pgo_nofuse.so   -DMV_PGO_FUSED_BUILD=0   every step builds afresh (A/B of the fused first trial, scripts/r4_pgo3_gpu.sh)
set -e
cd "$(dirname "$0")/../../mac-vo_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include"
OBJS=$(ls build/*.o | grep -v "build/pgo_solve.o")
hipcc $FL -DMV_PGO_STAMPS -c pgo_solve.hip -o /tmp/pgo_stamps.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/probes/libmacvo_hip_pgostamps.so $OBJS /tmp/pgo_stamps.o
hipcc $FL -DMV_PGO_FUSED_BUILD=0 -c pgo_solve.hip -o /tmp/pgo_nofuse.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/probes/libmacvo_hip_pgo_nofuse.so $OBJS /tmp/pgo_nofuse.o
ls -la ../../profiles/probes/libmacvo_hip_pgo*.so
