#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the secondary workloads (VQ-VAE, PM-VQVAE mnist / celeb_a, VDVAE) plus
# their throughput lines; run through gpurun from the repo root.  Output: gpurun_out/sec/*  (copy into profiles/).
set -e
R=/root/repo
O=$R/gpurun_out/sec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, script, args...
  name=$1; shift
  python3 "$@" > $O/${name}_bench.json 2> $O/${name}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 "$@" > $O/${name}_kt.log 2>&1
  cp $(ls $O/kt_$name/*/*kernel_stats.csv | tail -1) $O/${name}_kernel_stats.csv
  rm -rf $O/kt_$name
  echo "$name: $(tail -1 $O/${name}_bench.json | cut -c1-220)"
}
run vqvae_mnist $R/tools/bench_vqvae.py
run pm_vqvae_mnist $R/tools/bench_pm_vqvae.py --config mnist --steps 10 --warmup 3
run pm_vqvae_celeb_a $R/tools/bench_pm_vqvae.py --config celeb_a --steps 10 --warmup 3
run pm_vdvae_mnist $R/tools/bench_vdvae.py --batches 8,16 --steps 10 --warmup 3
