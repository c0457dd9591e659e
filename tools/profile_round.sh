#!/bin/bash
# rocprofv3 profiles of the bench step for profiles/<tag>_*: kernel statistics, then FETCH_SIZE and WRITE_SIZE in passes of their own
# usage on the GPU box: bash tools/profile_round.sh r02e   (then, back in the container: python profiles/summarize.py r02e)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
# the source these profiles are measured on (profiles/srchash.py): summarize.py writes it into the summary's header
python3 $R/profiles/srchash.py $R > $R/gpurun_out/prof_${TAG}_srchash.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-proof > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-proof > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-proof > $R/gpurun_out/prof_${TAG}_write.log 2>&1
tail -1 $R/gpurun_out/prof_${TAG}_stats.log | cut -c1-300
