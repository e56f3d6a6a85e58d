#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_fastmode.py tests/test_gpu_fullsize.py -q -x -k "16bit or stream or out16 or fast or volume or 720" 2>&1 | tail -4
python tools/kernel_bench.py volume_f16 --iters 60 2>&1 | grep -E "hwc(_out16)? B"
python tools/kernel_bench.py volume_f16 --iters 20 --H 720 --W 1280 2>&1 | grep -E "hwc(_out16)? B"
export MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_stamps.so
python profiles/probes/r6_hs_stamps.py out16 2 2>&1 | tail -7
