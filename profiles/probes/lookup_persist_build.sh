#!/bin/bash
# builds liblookup_persist_probe.so next to this script (build container; the .so is git-ignored and travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I../../mac-vo_amd/csrc -shared -o liblookup_persist_probe.so lookup_persist_probe.hip
ls -la liblookup_persist_probe.so
