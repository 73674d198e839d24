#!/bin/bash
# Timeline of the CONCURRENT step (two streams, launch plan): rocprofv3 --kernel-trace of bench.py, then
# tools/analyze_trace.py: per-queue busy time, kernels-in-flight histogram, timeline of the last step.
set -e
R=/root/repo
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 30 --warmup 10 --profile-steps 0 > $O/kt.log 2>&1
cd $R
python3 tools/analyze_trace.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) 10 $O/timeline.txt > $O/concurrency.txt
rm -rf $O/kt
cat $O/concurrency.txt
