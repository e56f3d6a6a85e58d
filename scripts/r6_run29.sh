#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do timeout 300 python tools/end_to_end.py --frames 40 --warmup 10 2>&1 | grep '^{"end_to_end"' | python -c "
import json,sys
e=json.loads(sys.stdin.read())['end_to_end']
for k in ('hooked','unhooked'):
    h=e[k]; print(k, {x:h.get(x) for x in ('network_ms_per_pair_batch','ms_per_frame','ms_per_frame_median','ms_per_frame_min','ms_per_frame_max','fps')})
"; done
