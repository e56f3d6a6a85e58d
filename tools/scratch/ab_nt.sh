timeout 100 python tools/scratch/hvol_probe.py
MV_H_STREAM=0 timeout 100 python tools/scratch/hvol_probe.py
B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
echo PLAIN; timeout 200 $B | tail -1
echo NT; MACVO_HIP_LIB=tools/scratch/libmacvo_hip_nt.so timeout 200 $B | tail -1
echo PLAIN-f16; timeout 200 $B --feat-dtype f16 --layout hwc | tail -1
echo NT-f16; MACVO_HIP_LIB=tools/scratch/libmacvo_hip_nt.so timeout 200 $B --feat-dtype f16 --layout hwc | tail -1
