mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r4/direct_test.log 2>&1; tail -3 gpurun_out/r4/direct_test.log
OLD=/root/repo/posterior_matching_amd/lib/ab/libpmhip_olddirect.so
{ tools/ab_workload.sh pm_vqvae_celeb_a 16 "-" "PM_LIB_PATH=$OLD" "-" "PM_LIB_PATH=$OLD"
  tools/ab_workload.sh pm_vqvae_mnist 256 "-" "PM_LIB_PATH=$OLD" "-" "PM_LIB_PATH=$OLD"
  tools/ab_workload.sh pm_vdvae_mnist 8 "-" "PM_LIB_PATH=$OLD" "-" "PM_LIB_PATH=$OLD"
  tools/ab_workload.sh pm_vae_mnist 256 "-" "PM_LIB_PATH=$OLD" "-" "PM_LIB_PATH=$OLD"
  tools/ab_workload.sh pm_vae_gas 128 "-" "PM_LIB_PATH=$OLD"
  tools/ab_workload.sh vqvae_mnist 256 "-" "PM_LIB_PATH=$OLD"
} > gpurun_out/r4/ab_direct.txt 2>&1; cat gpurun_out/r4/ab_direct.txt
