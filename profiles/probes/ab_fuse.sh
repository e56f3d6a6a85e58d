B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
timeout 300 python -m pytest tests/test_gpu_backend.py tests/test_gpu_native.py tests/test_gpu_pipeline.py tests/test_gpu_lanes.py -x -q 2>&1 | tail -3
echo FUSED; timeout 200 $B | tail -1
echo SEPARATE; MV_PIPE_FUSE_EPI=0 timeout 200 $B | tail -1
echo FUSED; timeout 200 $B | tail -1
echo SEPARATE; MV_PIPE_FUSE_EPI=0 timeout 200 $B | tail -1
echo FUSED-K20; timeout 200 $B --steps 20 --warmup 5 | tail -1
