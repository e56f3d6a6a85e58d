#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 0 1 2 3; do echo "== dd1 k=$k"; timeout 300 python profiles/probes/r6_queue_history.py $k 2>&1 | grep -v Warning | tail -6; done
echo "== dd1 k=1 with counts read-back"; PROBE_COUNTS=1 timeout 300 python profiles/probes/r6_queue_history.py 1 2>&1 | tail -3
for k in 0 1 2; do echo "== dd0 k=$k"; MV_PIPE_DEVICE_DRAW=0 timeout 300 python profiles/probes/r6_queue_history.py $k 2>&1 | tail -3; done
