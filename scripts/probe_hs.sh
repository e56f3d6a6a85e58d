#!/bin/bash
# Probes of the 16-bit streaming volume kernel (corr_volume_h_stream; round 6; results: profiles/r06_device_draw_ab.log, r06_hs_stamps.log).  Timing only: the knock-out /
# slack builds give wrong results by design.  Build first (container):
#   for s in 8 16 24; do scripts/probe_build.sh sl$s corr_volume.hip -DMV_HS_PROBE_SLACK=$s; done
#   for v in NOSTORE NODMA NOLDS NOMFMA; do scripts/probe_build.sh $v corr_volume.hip -DMV_HS_PROBE_$v; done
#   scripts/probe_build.sh stamps corr_volume.hip -DMV_HS_STAMPS
# then: gpurun -- 'bash scripts/probe_hs.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=$PWD/mac-vo_amd/csrc/build_probe
bench() { python tools/kernel_bench.py volume_f16 --iters 60 "$@" 2>&1 | grep -E "hwc(_out16)? B"; }
echo "== product"; bench
for v in sl8 sl16 sl24 NOSTORE NODMA NOLDS NOMFMA; do [ -f $P/libprobe_$v.so ] && { echo "== $v"; MACVO_HIP_LIB=$P/libprobe_$v.so bench; }; done
echo "== product 720p"; bench --H 720 --W 1280
echo "== one workgroup per CU / 1 and 4 column regions"; MV_H_STREAM_WGS=1 bench; MV_H_STREAM_REGIONS=1 bench; MV_H_STREAM_REGIONS=4 bench
for B in 8 32; do echo "== B=$B"; bench --B $B; done
if [ -f $P/libprobe_stamps.so ]; then
  export MACVO_HIP_LIB=$P/libprobe_stamps.so
  python profiles/probes/r6_hs_stamps.py out16 2 2>&1 | tail -7; python profiles/probes/r6_hs_stamps.py fp32 2 2>&1 | tail -7; python profiles/probes/r6_hs_stamps.py out16 16 2>&1 | tail -7
fi
