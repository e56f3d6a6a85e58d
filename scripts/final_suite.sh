#!/bin/bash
# the whole GPU suite + smoke() of the current tree (what the round-end driver runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T=${1:-suite}
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${T}_suite.log; tail -3 gpurun_out/${T}_suite.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-160 | tee gpurun_out/${T}_smoke.log
