#!/bin/bash
# SQ counters of the MSM kernels on uniformly random (dense) scalars: gpurun -- 'bash tools/sq_dense_sort.sh'
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/sqd -o sq -- python3 $R/tools/msm_ntt_micro.py 1024 > $R/gpurun_out/sqd.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/sqd/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"]
print("kernel,launches,ms," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:6]:
    print(f"{k},{n[k]},{dur[k]:.2f}," + ",".join(f"{v.get(x, 0):.3g}" for x in names))
PY
