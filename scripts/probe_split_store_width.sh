#!/bin/bash
# f16x2 GEMM: do its stores cost per instruction or per byte?  (round 6; profiles/r06_device_draw_ab.log)  Build first (container):
#   scripts/probe_build.sh STORE4 corr_volume_split.hip -DMV_SPLIT_PROBE_STORE4; scripts/probe_build.sh STORE1OF4 corr_volume_split.hip -DMV_SPLIT_PROBE_STORE1OF4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MV_SPLIT_MODE=f16x2
P=$PWD/mac-vo_amd/csrc/build_probe
for rep in 1 2; do
echo "== product"; python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
echo "== 8 x dwordx4 stores instead of 32 x dword (same bytes, garbage values)"; MACVO_HIP_LIB=$P/libprobe_STORE4.so python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
echo "== 8 x dword stores (a quarter of the bytes)"; MACVO_HIP_LIB=$P/libprobe_STORE1OF4.so python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
done
