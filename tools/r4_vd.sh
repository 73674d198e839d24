mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_vdvae.py -x -q > gpurun_out/r4/quick_test.log 2>&1; tail -3 gpurun_out/r4/quick_test.log
{ tools/ab_workload.sh pm_vdvae_mnist 8 "-" "-" "-"
  tools/ab_workload.sh pm_vdvae_mnist 16 "-" "-"; } > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
