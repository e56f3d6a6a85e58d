#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06f_suite.log; tail -3 gpurun_out/r06f_suite.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
