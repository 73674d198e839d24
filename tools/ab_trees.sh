#!/bin/bash
# same-box comparison of source trees: tools/ab_trees.sh <rounds> <dir> <dir> ...   (each tree with its own built libpmhip.so)
R=$1; shift
for i in $(seq $R); do
  for d in "$@"; do
    r=$(cd $d && python bench.py --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --profile-steps 0 --spread-steps 0 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$d => $r"
  done
done
