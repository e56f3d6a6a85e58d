B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo DMA3; timeout 200 $B | tail -1
echo DMA2_BK32; MV_VOL_DMA=2 timeout 200 $B | tail -1
echo DMA3; timeout 200 $B | tail -1
echo DMA2_BK32; MV_VOL_DMA=2 timeout 200 $B | tail -1
echo DMA2_BK32-l3; MV_VOL_DMA=2 timeout 200 $B --lanes 3 | tail -1
echo DMA3-l3; timeout 200 $B --lanes 3 | tail -1
echo DMA2_BK32-K20; MV_VOL_DMA=2 timeout 200 $B --steps 20 --warmup 5 | tail -1
