#!/bin/bash
# round 4: partial sums vs atomics, eager vs tail reduction (same box)
cd /root/repo
O=gpurun_out/r4
mkdir -p $O
bash tools/ab.sh - "PM_PART_EAGER_MB=100000" "PM_NO_PARTIALS=1" - "PM_PART_EAGER_MB=100000" "PM_NO_PARTIALS=1" "PM_PART_EAGER_MB=12" > $O/ab1.txt 2>&1
for v in default nopart; do
  e=""; [ $v = nopart ] && e="PM_NO_PARTIALS=1"
  env $e PM_BENCH_KERNEL_TABLE=$O/ktable_$v.txt python bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 20 --warmup 5 > $O/serial_$v.json 2> $O/serial_$v.err
done
cat $O/ab1.txt
