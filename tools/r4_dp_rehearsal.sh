#!/bin/bash
# `bench.py --gpus 2 --backend gloo` rehearsal of the three workloads BASELINE.json calls multi-GPU, two ranks
# sharing the one GPU of the box (weak and strong scaling forms); JSON lines -> profiles/r04_bench_gpus2_gloo_<workload>.json
cd /root/repo
O=gpurun_out/r4
mkdir -p $O
for w in pm_vae_mnist pm_vdvae_mnist pm_vqvae_celeb_a; do
  for sc in weak strong; do
    [ $w = pm_vae_mnist ] && [ $sc = weak ] && extra="--no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --profile-steps 0 --spread-steps 0" || extra=""
    timeout -k 10 240 python bench.py --gpus 2 --backend gloo --workload $w --scaling $sc --steps 6 --warmup 3 $extra > $O/gloo2_${w}_${sc}.json 2> $O/gloo2_${w}_${sc}.err
    echo "$w $sc rc=$? $(cut -c1-260 $O/gloo2_${w}_${sc}.json)"
  done
done
