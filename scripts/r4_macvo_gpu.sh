#!/bin/bash
# Round-4, first GPU call: the reference's own MACVO loop with the HIP plugins (tests/test_macvo_run.py), then the whole GPU suite and a bench line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r04_macvo.log; : > $L
timeout 900 python -m pytest tests/test_macvo_run.py -m gpu -x -q 2>&1 | tail -25 >> $L
for c in tartan_fast synth_fast; do
  timeout 300 python tests/refrun.py --mode ref --case $c --frames 6 2>&1 | grep '^{' >> $L
  timeout 300 python tests/refrun.py --mode hip --case $c --frames 6 2>&1 | grep '^{' >> $L
done
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> $L
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r04_bench_first_line.json
cat $L
