mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_vdvae.py tests/test_gpu_repeatability.py -x -q -k "vdvae" > gpurun_out/r4/quick_test.log 2>&1; tail -3 gpurun_out/r4/quick_test.log
{ tools/ab_workload.sh pm_vdvae_mnist 8 "-" "PM_VDVAE_NO_SAMPLE_PROJECT=1" "-" "PM_VDVAE_NO_SAMPLE_PROJECT=1"; tools/ab_workload.sh pm_vdvae_mnist 16 "-" "PM_VDVAE_NO_SAMPLE_PROJECT=1"; } > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
python tools/stamp_timeline.py gpurun_out/r4/stamps_vdvae_sp.txt pm_vdvae_mnist > /dev/null 2>&1; tail -1 gpurun_out/r4/stamps_vdvae_sp.txt
