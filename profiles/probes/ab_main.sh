B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo MAIN2; MV_PIPE_MAIN_STREAMS=2 timeout 200 $B | tail -1
echo MAIN1; timeout 200 $B | tail -1
echo MAIN2; MV_PIPE_MAIN_STREAMS=2 timeout 200 $B | tail -1
echo MAIN2-K20; MV_PIPE_MAIN_STREAMS=2 timeout 200 $B --steps 20 --warmup 5 | tail -1
echo MAIN2-l3; MV_PIPE_MAIN_STREAMS=2 timeout 200 $B --lanes 3 | tail -1
