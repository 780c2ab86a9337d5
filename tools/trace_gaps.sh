#!/bin/bash
# kernel-trace of a short graph-replay bench, kept as a trace (not just stats): tools/trace_gaps.sh <tag> [from_us to_us] -> gpurun_out/trace_<tag>/ (+ every kernel of that window of the step)
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --no-spawn --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-extra-configs > $R/gpurun_out/trace_$TAG.log 2>&1
python3 $R/tools/trace_gaps.py $R/gpurun_out/trace_$TAG > $R/gpurun_out/trace_$TAG.txt 2>&1
python3 $R/tools/trace_timeline.py $R/gpurun_out/trace_$TAG $2 $3 > $R/gpurun_out/timeline_$TAG.txt 2>&1
find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" -delete
tail -60 $R/gpurun_out/trace_$TAG.txt
