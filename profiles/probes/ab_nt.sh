B="python bench.py --no-cpu-baseline --config4-steps 0 --no-decoder-leg"
for i in 1 2; do
echo DEFAULT_NT; timeout 200 $B | tail -1
echo PLAIN_LIB; MACVO_HIP_LIB=profiles/probes/libmacvo_hip_plain.so timeout 200 $B | tail -1
done
echo PLAIN-f16; timeout 200 $B --feat-dtype f16 --layout hwc | tail -1
echo PLAIN_LIB-f16; MACVO_HIP_LIB=profiles/probes/libmacvo_hip_plain.so timeout 200 $B --feat-dtype f16 --layout hwc | tail -1
echo PLAIN-l3; timeout 200 $B --lanes 3 | tail -1
echo PLAIN_LIB-l3; MACVO_HIP_LIB=profiles/probes/libmacvo_hip_plain.so timeout 200 $B --lanes 3 | tail -1
