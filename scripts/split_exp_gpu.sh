cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r3_split5.log
: > $L
python tools/kernel_bench.py volume_split --iters 50 2>&1 | grep "volume_split " >> $L
for k in 0 1 2 4 8 3 9 15; do echo "== compile-time knock-out $k (1 stores, 2 DMA, 4 barrier, 8 LDS reads)" >> $L; MACVO_HIP_LIB=$PWD/tools/scratch/libmacvo_hip_split_k$k.so python tools/kernel_bench.py volume_split --iters 50 2>&1 | grep "volume_split " >> $L; done
cat $L
