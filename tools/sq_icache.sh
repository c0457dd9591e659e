#!/bin/bash
# instruction-cache requests / misses per kernel of one bench step (rocprofv3 SQC counters): gpurun -- 'bash tools/sq_icache.sh'
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/sqi -o sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-proof > $R/gpurun_out/sqi.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/sqi/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
names = ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"]
print("kernel," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:8]:
    print(k + "," + ",".join(f"{v.get(n, 0):.0f}" for n in names))
PY
