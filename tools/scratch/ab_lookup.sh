B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo DEFAULT; timeout 200 $B | tail -1
echo QPB8; MV_LOOKUP_QPB=8 timeout 200 $B | tail -1
echo SMALL0; MV_LOOKUP_SMALL=0 timeout 200 $B | tail -1
