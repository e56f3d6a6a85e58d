cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp MV_SPLIT_MODE=f16x2
L=gpurun_out/r3_split9.log
: > $L
for w in 4 8; do
python tools/kernel_bench.py volume_split --iters 50 2>&1 | grep "volume_split " | sed "s/^/waves=$w product /" >> $L
MV_SPLIT_WAVES=$w python tools/kernel_bench.py volume_split --iters 50 --zeros 2>&1 | grep "volume_split " | sed "s/^/waves=$w product /" >> $L
for k in 0 1 2 8 4 15; do MV_SPLIT_WAVES=$w MACVO_HIP_LIB=$PWD/profiles/probes/libmacvo_hip_split_k$k.so python tools/kernel_bench.py volume_split --iters 50 2>&1 | grep "volume_split " | sed "s/^/waves=$w knock=$k (1 stores 2 DMA 4 barrier 8 LDS reads) /" >> $L; done
done
cat $L
