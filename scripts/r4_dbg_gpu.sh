cd $GRAFT_REPO_ROOT
MV_PIPE_LOOKUPS_ON=vol timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-steps 0 --config4-steps 0 --no-decoder-leg 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-600
