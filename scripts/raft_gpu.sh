#!/bin/bash
# Runs HERE (build container): ships the reference's Python tree (Module/, Utility/, DataLoader/ — .py files only, ~118 KB as xz + base64) INSIDE the gpurun
# command line — the sources are not copied into this repository — and runs tools/raft_gpu_probe.py on the MI355X box.
set -e
cd /root/reference
B64=$(tar cf - $(find Module Utility DataLoader -name "*.py") | xz -9e -c | base64 -w0)
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p /tmp/ref_py && echo $B64 | base64 -d | xz -d | tar x -C /tmp/ref_py && MACVO_REFERENCE_ROOT=/tmp/ref_py python tools/raft_gpu_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_raft_gpu.log"
