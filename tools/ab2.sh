#!/bin/bash
# A/B runs of a secondary workload on ONE box: tools/ab2.sh "<workload> [batch]" <env settings>...   ("-" = defaults)
cd /root/repo
w="$1"; shift
for v in "$@"; do
  envs=$(echo "$v" | tr ';' ' ')
  [ "$v" = "-" ] && envs=""
  r=$(env $envs python tools/measure.py $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['images_per_sec'], d['ms_per_step'], d['loss'])")
  echo "$w | $v => $r"
done
