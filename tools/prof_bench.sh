#!/bin/bash
# kernel-trace profile of the default bench (graph replay): tools/prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-spawn --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_$TAG -name "*_stats.csv" | head
tail -1 $R/gpurun_out/prof_$TAG.log
