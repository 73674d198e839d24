#!/bin/bash
# A/B runs of bench.py on ONE box: each argument is a ';'-separated list of VAR=value settings ("-" = defaults)
cd /root/repo
for v in "$@"; do
  envs=$(echo "$v" | tr ';' ' ')
  [ "$v" = "-" ] && envs=""
  r=$(env $envs python bench.py --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --profile-steps 0 --spread-steps 0 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['aux']['loss'])")
  echo "$v => $r"
done
