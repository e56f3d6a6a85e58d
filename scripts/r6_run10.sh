#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in threads8 oracle extras ops subprocess; do PROBE_MODE=$m timeout 300 python profiles/probes/r6_queue_history.py 0 1 2>&1 | grep "lane pipe" | tail -2; done
