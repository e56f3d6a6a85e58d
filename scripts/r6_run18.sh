#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== product"; python tools/kernel_bench.py volume_f16 --iters 60 2>&1 | grep -E "out16|hwc B"
for sl in 8 16 24; do echo "== slack $sl (timing probe, wrong results)"; MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_sl$sl.so python tools/kernel_bench.py volume_f16 --iters 60 2>&1 | grep -E "out16|hwc B"; done
echo "== product 720p"; python tools/kernel_bench.py volume_f16 --iters 30 --H 720 --W 1280 2>&1 | grep -E "out16|hwc B"
echo "== slack 24 720p"; MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_sl24.so python tools/kernel_bench.py volume_f16 --iters 30 --H 720 --W 1280 2>&1 | grep -E "out16|hwc B"
