#!/bin/bash
# A/B of the fixed-order forms against the atomic forms they replace, one box: tools/r4_ab_det.sh > profiles/r04_ab_determinism.txt
cd /root/repo
for rep in 1 2; do
tools/ab_workload.sh vqvae_mnist 256 "-" "PM_VQ_DW_ATOMIC=1"
tools/ab_workload.sh pm_vqvae_celeb_a 16 "-" "PM_SPLITK_ATOMIC=1"
tools/ab_workload.sh pm_vae_gas 128 "-" "PM_SPLITK_ATOMIC=1;PM_NLL_ATOMIC=1"
tools/ab_workload.sh pm_vqvae_mnist 32 "-" "PM_SPLITK_ATOMIC=1"
done
