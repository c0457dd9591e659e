#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_rNN_{stats,fetch,write}) into the small
tracked summaries under profiles/.  Usage: python profiles/summarize.py r01 [suffix]"""
import collections
import csv
import glob
import shutil
import sys

tag = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""
src = f"gpurun_out/prof_{tag}{suffix}"
stats = glob.glob(f"{src}_stats/*/*_kernel_stats.csv") + glob.glob(f"{src}_stats/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}{suffix}_kernel_stats.csv")
else:
    # rocprofv3 without --output-format csv leaves a rocpd SQLite database: the same per-kernel statistics from its `kernels` view
    import sqlite3
    con = sqlite3.connect(glob.glob(f"{src}_stats/*/*_results.db")[0])
    rows_k = con.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows_k)
    with open(f"profiles/{tag}{suffix}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows_k:
            w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100 * r[2] / tot, 2), r[4], r[5]])
rows = []
for kind in ("fetch", "write"):
    files = glob.glob(f"{src}_{kind}/*/*_counter_collection.csv") + glob.glob(f"{src}_{kind}/*counter_collection.csv")
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
        agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (n, v, ms) in agg.items():
        rows.append((kind, k, n, v, ms))
# the source the counters were measured on (tools/profile_round.sh records it on the GPU box, from the snapshot it profiled):
# kernel -> SHA-256 of its .hip file and the headers that file includes (profiles/srchash.py); bench.py refuses a summary whose
# hash for the dominant kernel is not the running tree's
import json
import os
hashes = {}
if os.path.exists(f"{src}_srchash.json"):
    hashes = json.load(open(f"{src}_srchash.json"))
with open(f"profiles/{tag}{suffix}_pmc_summary.csv", "w") as f:
    if hashes:
        kernels = sorted({k.replace("void ", "").replace("vdb::", "").split("<")[0] for _kind, k, *_ in rows})
        f.write("# source_sha256 " + " ".join(f"{k}={hashes[k]}" for k in kernels if k in hashes) + "\n")
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-proof (tools/profile_round.sh)\n")
    f.write("# Counter_Value is in KiB as reported; gfx950 correction (MI355X_MICROARCH.md §HBM): HBM read bytes = 2 * FETCH_SIZE * 1024\n")
    f.write("# for wide coalesced 16 B/lane streams, HBM write bytes = WRITE_SIZE * 1024.\n")
    w = csv.writer(f)       # (kernel names of templates hold commas: quoted)
    w.writerow(["counter", "kernel", "launches", "sum_counter_KiB", "per_launch_KiB", "sum_ms"])
    for kind, k, n, v, ms in sorted(rows, key=lambda r: (r[0], -r[3])):
        w.writerow([kind, k, n, f"{v:.1f}", f"{v / n:.1f}", f"{ms:.3f}"])
print("wrote", f"profiles/{tag}{suffix}_kernel_stats.csv", f"profiles/{tag}{suffix}_pmc_summary.csv")
