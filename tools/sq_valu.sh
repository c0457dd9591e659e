#!/bin/bash
# VALU instruction count per kernel of one bench step (rocprofv3 SQ counters): gpurun -- 'bash tools/sq_valu.sh'
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/sqv -o sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-proof > $R/gpurun_out/sqv.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/sqv/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
out = open("$R/gpurun_out/sq_valu_summary.csv", "w", newline="")
w = csv.writer(out)      # (template kernels' names hold commas: quoted)
w.writerow(["kernel", "launches", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:16]:
    w.writerow([k, n[k], f"{v.get('SQ_INSTS_VALU',0):.0f}", f"{v.get('SQ_ACTIVE_INST_VALU',0):.0f}", f"{v.get('GRBM_GUI_ACTIVE',0):.0f}"])
out.close()
print(open("$R/gpurun_out/sq_valu_summary.csv").read())
PY
