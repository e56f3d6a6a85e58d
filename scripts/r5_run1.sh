set -x
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_corr.py -q -m gpu -k "upsample" 2>&1 | tail -5
for n in 4 8; do MV_UPS_NSX=$n timeout 120 python profiles/probes/r5_upsample_ab.py; done
timeout 120 python profiles/probes/r5_upsample_ab.py
