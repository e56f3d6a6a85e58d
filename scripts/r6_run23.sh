#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== product"; python tools/kernel_bench.py volume_f16 --iters 60 2>&1 | grep -E "float16_hwc"
for v in NOSTORE NODMA NOLDS NOMFMA; do echo "== $v (timing probe, wrong results)"; MACVO_HIP_LIB=$PWD/mac-vo_amd/csrc/build_probe/libprobe_$v.so python tools/kernel_bench.py volume_f16 --iters 60 2>&1 | grep -E "float16_hwc"; done
