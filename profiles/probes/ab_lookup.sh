B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
MV_LOOKUP_QPB=1 timeout 200 python -m pytest tests/test_gpu_corr.py -q -x -k lookup 2>&1 | tail -2
echo DEFAULT; timeout 200 $B | tail -1
echo QPW1; MV_LOOKUP_QPB=1 timeout 200 $B | tail -1
echo DEFAULT; timeout 200 $B | tail -1
echo QPW1; MV_LOOKUP_QPB=1 timeout 200 $B | tail -1
