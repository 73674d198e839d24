mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gmm or model_forward or gradients_at or repeat" > gpurun_out/r4/quick_test.log 2>&1; tail -3 gpurun_out/r4/quick_test.log
{ tools/ab_workload.sh pm_vae_mnist 256 "-" "-" "-"; } > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
python tools/stamp_timeline.py gpurun_out/r4/stamps_pm_vae3.txt pm_vae_mnist > /dev/null 2>&1; grep "gmm_logprob\|span" gpurun_out/r4/stamps_pm_vae3.txt
