#!/bin/bash
# SQ counters per kernel of a workload's step (one PMC pass, counters only: no trace domains beside them):
#   tools/pmc_sq.sh <workload> <out-prefix>   ->  gpurun_out/<out-prefix>_sq.txt
set -e
R=/root/repo
W=$1
O=$R/gpurun_out/$2
mkdir -p $(dirname $O)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq_$$
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq_$$ -- python3 $R/tools/run_steps.py $W 3 > ${O}_sq.log 2>&1
cd $R
python3 profiles/summarize_sq.py $(ls /tmp/pmc_sq_$$/*/*counter_collection.csv | tail -1) > ${O}_sq.txt
rm -rf /tmp/pmc_sq_$$
head -30 ${O}_sq.txt
