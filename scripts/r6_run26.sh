#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_stamps.so
python profiles/probes/r6_hs_stamps.py out16 2 2>&1 | grep -v Warn | tail -8
python profiles/probes/r6_hs_stamps.py fp32 2 2>&1 | tail -8
python profiles/probes/r6_hs_stamps.py out16 16 2>&1 | tail -8
