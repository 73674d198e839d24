#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters in KB).

    python profiles/make_pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix>

Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 tallies the 128-B requests of wide (16 B/lane) coalesced
loads at 64 B: traffic = 2 x FETCH_SIZE + WRITE_SIZE (every GEMM-class kernel here loads b128); WRITE_SIZE is exact for
16-B stores and f32 atomics.  Writes <out_prefix>.txt (table) and <out_prefix>.json (read by bench.py).
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if ">(" in name:
        return name.split(">(")[0] + ">"
    return name.split("(")[0]


def load(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = short(r["Kernel_Name"])
        tot[n] += float(r["Counter_Value"])
        disp[n].add(r["Dispatch_Id"])
    return tot, disp


fetch, fd = load(sys.argv[1], "FETCH_SIZE")
write, wd = load(sys.argv[2], "WRITE_SIZE")
rows = {}
for n in fetch:
    launches = max(len(fd[n]), 1)
    f = fetch[n] / launches
    w = write.get(n, 0.0) / max(len(wd.get(n, ())), 1)
    rows[n] = {"launches": launches, "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
               "traffic_bytes": int((2 * f + w) * 1024)}
order = sorted(rows, key=lambda n: -rows[n]["traffic_bytes"] * rows[n]["launches"])
with open(sys.argv[3] + ".txt", "w") as fp:
    fp.write(__doc__.split("\n\n")[0].strip() + "\n# traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md)\n")
    fp.write(f"{'kernel':58s} {'launches':>8s} {'FETCH_SIZE KB':>14s} {'WRITE_SIZE KB':>14s} {'traffic MB (2F+W)':>18s}\n")
    for n in order:
        r = rows[n]
        fp.write(f"{n[:58]:58s} {r['launches']:8d} {r['fetch_size_kb']:14.1f} {r['write_size_kb']:14.1f} "
                 f"{r['traffic_bytes'] / 1048576:18.2f}\n")
json.dump({n: rows[n] for n in order}, open(sys.argv[3] + ".json", "w"), indent=1)
