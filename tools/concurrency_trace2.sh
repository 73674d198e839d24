#!/bin/bash
# Timeline of a secondary workload's step: tools/concurrency_trace2.sh <out dir name> <workload> [batch]
set -e
R=/root/repo
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/measure.py $2 ${3:--} 12 4 > $O/kt.log 2>&1
cd $R
python3 tools/analyze_trace.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) 6 $O/timeline.txt > $O/concurrency.txt
rm -rf $O/kt
cat $O/concurrency.txt
