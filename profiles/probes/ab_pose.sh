B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo SPLIT-l3; timeout 200 $B --lanes 3 | tail -1
echo OLD-l3; MV_PIPE_POSE_SPLIT=0 timeout 200 $B --lanes 3 | tail -1
echo SPLIT; timeout 200 $B | tail -1
echo OLD; MV_PIPE_POSE_SPLIT=0 timeout 200 $B | tail -1
echo SPLIT-l3; timeout 200 $B --lanes 3 | tail -1
echo OLD-l3; MV_PIPE_POSE_SPLIT=0 timeout 200 $B --lanes 3 | tail -1
