#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_patch_embed.py -q -m gpu 2>&1 | tail -6
