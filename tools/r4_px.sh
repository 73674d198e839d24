mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_pixelcnn.py tests/test_gpu_celeba.py -x -q > gpurun_out/r4/quick_test.log 2>&1; tail -3 gpurun_out/r4/quick_test.log
{ tools/ab_workload.sh pm_vqvae_celeb_a 16 "-" "PM_SKINNY_MAXN=1000000" "-" "PM_SKINNY_MAXN=1000000"
  tools/ab_workload.sh pm_vqvae_mnist 256 "-" "PM_SKINNY_MAXN=1000000" "-" "PM_SKINNY_MAXN=1000000"; } > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
