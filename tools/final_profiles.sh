#!/bin/bash
# Regenerates the judged profiles of a round on the GPU box (run through gpurun from the repo root):
#   bench line + live kernel table, rocprofv3 kernel stats of the serial run, two PMC passes (FETCH_SIZE / WRITE_SIZE).
set -e
R=/root/repo
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PM_BENCH_KERNEL_TABLE=$O/bench_kernel_table.txt python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 > $O/bench_serial.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 50 --warmup 10 --profile-steps 0 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 6 --warmup 2 --profile-steps 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 6 --warmup 2 --profile-steps 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --serial --no-cpu-baseline --no-pmc --no-f32-aux --no-secondary --spread-steps 0 --steps 6 --warmup 2 --profile-steps 0 > $O/pmc_mfma.log 2>&1
cd $R
python3 profiles/make_pmc_mfma.py $(ls $O/pmc_mfma/*/*counter_collection.csv | tail -1) $O/pmc_mfma.txt
rm -rf $O/pmc_mfma
python3 profiles/make_pmc_traffic.py $(ls $O/pmc_fetch/*/*counter_collection.csv | tail -1) $(ls $O/pmc_write/*/*counter_collection.csv | tail -1) $O/pmc_traffic
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/serial_kernel_stats.csv
rm -rf $O/kt/*/*kernel_trace.csv $O/pmc_fetch $O/pmc_write     # large per-dispatch files stay on the box
tail -1 $O/bench.json | cut -c1-200
