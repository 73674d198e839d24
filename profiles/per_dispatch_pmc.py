#!/usr/bin/env python3
"""Per-kernel (last dispatch of each name+grid) MFMA utilisation / occupancy / wait shares from a
rocprofv3 --pmc counter_collection CSV collected with the SQ_* set used in DESIGN.md."""
import csv
import sys
from collections import OrderedDict

rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]
    rows.setdefault(d, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size")})
    rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
seen = OrderedDict()
for v in rows.values():
    if "gemm" not in v["name"] and "wgrad" not in v["name"]:
        continue
    name = v["name"].replace("void (anonymous namespace)::", "").split("(")[0] + " wgs=" + str(int(v["grid"]) // 256)
    seen[name] = v
for name, v in seen.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    wc = v["SQ_WAVE_CYCLES"]
    print(f"{name:50s} cyc {cyc:8.0f} mfma_util {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):4.2f} "
          f"waves/simd {wc * 4 / (cyc * 1024):4.1f} wait_inst {v['SQ_WAIT_INST_ANY'] / wc:4.2f} "
          f"wait_any {v['SQ_WAIT_ANY'] / wc:4.2f} active {v['SQ_ACTIVE_INST_ANY'] / wc:4.2f} "
          f"valu {v['SQ_ACTIVE_INST_VALU'] / wc:4.2f} lds_conf {v['SQ_LDS_BANK_CONFLICT'] / wc:5.3f}")
