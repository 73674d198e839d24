mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repeatability.py tests/test_gpu_zz_coverage.py -x -q -k "model or train or gradients or repeat or reproducible or golden or coverage or batch" > gpurun_out/r4/quick_test.log 2>&1; tail -3 gpurun_out/r4/quick_test.log
{ tools/ab_workload.sh pm_vae_mnist 256 "-" "PM_NO_COLSUM_MULTI=1" "-" "PM_NO_COLSUM_MULTI=1" "-" "PM_NO_COLSUM_MULTI=1"; } > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
