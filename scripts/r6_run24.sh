#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== B=2 default"; python tools/kernel_bench.py volume_f16 --iters 40 2>&1 | grep -E "hwc(_out16)? B"
echo "== B=2 WGS=1"; MV_H_STREAM_WGS=1 python tools/kernel_bench.py volume_f16 --iters 40 2>&1 | grep -E "hwc(_out16)? B"
echo "== B=2 regions 1"; MV_H_STREAM_REGIONS=1 python tools/kernel_bench.py volume_f16 --iters 40 2>&1 | grep -E "hwc(_out16)? B"
echo "== B=2 regions 4"; MV_H_STREAM_REGIONS=4 python tools/kernel_bench.py volume_f16 --iters 40 2>&1 | grep -E "hwc(_out16)? B"
echo "== B=8"; python tools/kernel_bench.py volume_f16 --iters 20 --B 8 2>&1 | grep -E "hwc(_out16)? B"
echo "== B=32"; python tools/kernel_bench.py volume_f16 --iters 10 --B 32 2>&1 | grep -E "hwc(_out16)? B"
