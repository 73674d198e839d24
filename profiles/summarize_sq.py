#!/usr/bin/env python3
"""Per-kernel SQ picture from one rocprofv3 --pmc pass (tools/pmc_sq.sh): sums over dispatches, then ratios.
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)        share of matrix-pipe cycles busy
  waves/simd = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)                     (SQ_* wave counters are quad-cycles)
  wait_any / wait_inst / active / valu = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU over SQ_WAVE_CYCLES"""
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
print(f"{'kernel':46s} {'launches':>8s} {'us/launch@2.0GHz':>16s} {'mfma_busy':>9s} {'waves/simd':>10s} {'wait_any':>8s} {'wait_inst':>9s} "
      f"{'active':>6s} {'valu':>6s} {'lds_conf':>8s}")
for n in sorted(rows, key=lambda k: -rows[k].get("GRBM_GUI_ACTIVE", 0.0)):
    c = rows[n]
    act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if act <= 0 or wc <= 0:
        continue
    L = len(calls[n])
    print(f"{n[:46]:46s} {L:8d} {act / L / 2000.0:16.1f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (act * 1024):9.3f} "
          f"{4 * wc / (act * 1024):10.2f} {c.get('SQ_WAIT_ANY', 0) / wc:8.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:9.2f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f} "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / (act * 256 * 4) if act else 0:8.3f}")
