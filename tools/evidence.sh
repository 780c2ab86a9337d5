#!/bin/bash
# Evidence run kept under profiles/ (copied from gpurun_out/): tools/evidence.sh <tag>
#   bench line, kernel stats of the default bench, PMC traffic, render pass (config 5) + its kernel stats, config-2 forward-only,
#   train sanity trace.
TAG=${1:-r02x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev_$TAG
mkdir -p $O
# the two --pmc passes come first: bench.py reads profiles/r06_pmc_traffic.json for roofline.traffic (and checks it against the kernel
# sources), so the committed bench line and the counters describe the same binaries
(cd /tmp && export TMPDIR=/tmp && for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$c -- python3 $R/bench.py --no-spawn --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-exact-f32 --no-extra-configs --no-live-pmc > $R/gpurun_out/pmc_bench_$c.log 2>&1
done)
python3 $R/tools/pmc_step_traffic.py 0 r06_pmc_traffic.json > $O/pmc_traffic.txt 2>&1; cp $R/profiles/r06_pmc_traffic.json $O/ 2>/dev/null
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_line.json 2> $O/bench.err
tail -1 $O/bench_line.json
python tools/bench_forward_only.py > $O/forward_only.json 2> $O/forward_only.err; tail -1 $O/forward_only.json
python tools/bench_render.py > $O/bench_render.log 2>&1; tail -2 $O/bench_render.log
python tools/train_sanity.py 300 > $O/train_sanity.log 2>&1; tail -3 $O/train_sanity.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-spawn --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-extra-configs > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_render -- python3 $R/tools/bench_render.py 270 480 > $O/prof_render.log 2>&1
find $O -name "*kernel_trace.csv" -delete
bash $R/tools/trace_gaps.sh $TAG > /dev/null 2>&1; cp $R/gpurun_out/timeline_$TAG.txt $R/gpurun_out/trace_$TAG.txt $O/ 2>/dev/null
find $O -name "*_stats.csv" | head
