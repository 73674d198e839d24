#!/bin/bash
# A/B runs of one secondary workload on ONE box: tools/ab_workload.sh <workload> <batch> "VAR=v;VAR2=w" "-" ...
cd /root/repo
w=$1; b=$2; shift 2
for v in "$@"; do
  envs=$(echo "$v" | tr ';' ' ')
  [ "$v" = "-" ] && envs=""
  r=$(env $envs python - <<PY 2>/dev/null
import sys
sys.path.insert(0, "/root/repo")
from tools.workloads import build, measure
w = build("$w", $b)
m = measure(w, 40, 8)
print(m["images_per_sec"], m["ms_per_step"], m["loss"])
PY
)
  echo "$w B=$b $v => $r"
done
