#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --serial ...
    python3 profiles/make_pmc_mfma.py <counter_collection.csv> <out.txt> [algorithmic GFLOP per launch json]
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles summed over the chip's 1024 SIMD matrix pipes
against the kernel's active cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  For v_mfma_f32_32x32x16_bf16 (32 cycles each)
a value of 1.0 would be the dense bf16 peak; a bf16x3 kernel issues three such products per algorithmic MAC."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return (name.split(">(")[0] + ">") if ">(" in name else name.split("(")[0]


rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        n = short(r["Kernel_Name"])
        rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[n].add(r["Dispatch_Id"])
out = open(sys.argv[2], "w")
out.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --serial (program directly after --)\n")
out.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); cycles/launch = GRBM_GUI_ACTIVE / 8 / launches\n")
out.write(f"{'kernel':48s} {'launches':>8s} {'cycles/launch':>14s} {'MFMA busy cyc/launch':>22s} {'mfma_busy':>10s} {'SQ_BUSY_CYCLES/launch':>22s}\n")
order = sorted(rows, key=lambda n: -rows[n].get("GRBM_GUI_ACTIVE", 0.0))
for n in order:
    c = rows[n]
    L = max(len(calls[n]), 1)
    act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if act <= 0:
        continue
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    sqb = c.get("SQ_BUSY_CYCLES", 0.0)
    out.write(f"{n[:48]:48s} {L:8d} {act / L:14.0f} {busy / L:22.0f} {busy / (act * 1024.0):10.4f} {sqb / L:22.0f}\n")
out.close()
