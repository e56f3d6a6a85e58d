B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo BK32; timeout 200 $B | tail -1
echo BK16; MV_VOL_DMA=16 timeout 200 $B | tail -1
echo BK32; timeout 200 $B | tail -1
echo BK16; MV_VOL_DMA=16 timeout 200 $B | tail -1
