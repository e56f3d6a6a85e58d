#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MV_SPLIT_MODE=f16x2
for rep in 1 2; do
echo "== product"; python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
echo "== 8 x dwordx4 stores instead of 32 x dword (same bytes, garbage values)"; MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_STORE4.so python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
echo "== 8 x dword stores (a quarter of the bytes)"; MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_STORE1OF4.so python tools/kernel_bench.py volume_split --iters 80 2>&1 | grep "volume_split"
done
