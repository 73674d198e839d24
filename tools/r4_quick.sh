mkdir -p gpurun_out/r4
P=/root/repo/posterior_matching_amd/lib/ab
tools/ab_workload.sh pm_vdvae_mnist 8 "-" "PM_LIB_PATH=$P/libpmhip_spnw8.so" "PM_LIB_PATH=$P/libpmhip_spnw4.so" "-" "PM_LIB_PATH=$P/libpmhip_spnw8.so" "PM_LIB_PATH=$P/libpmhip_spnw4.so" > gpurun_out/r4/ab_quick.txt 2>&1; cat gpurun_out/r4/ab_quick.txt
