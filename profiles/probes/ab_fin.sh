B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo DEFAULT; timeout 200 $B | tail -1
echo FIN512; MV_KP_FINISH_SMALL_NT=512 timeout 200 $B | tail -1
echo DEFAULT; timeout 200 $B | tail -1
echo FIN512; MV_KP_FINISH_SMALL_NT=512 timeout 200 $B | tail -1
