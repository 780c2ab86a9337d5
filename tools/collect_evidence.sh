#!/bin/bash
# copy the summaries of an evidence run (tools/evidence.sh <tag> on the GPU box, merged back under gpurun_out/ev_<tag>) into profiles/
TAG=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/ev_$TAG
P=$R/profiles
cp $O/bench_line.json $P/${TAG}_bench_line.json
cp $O/forward_only.json $P/${TAG}_forward_only_config2.json
cp $O/bench_render.log $P/${TAG}_bench_render_config5.log
cp $O/pmc_traffic.txt $P/${TAG}_pmc_traffic.txt
cp $O/smoke.log $P/${TAG}_smoke.log
cp $O/train_sanity.log $P/${TAG}_train_sanity.log
cp $O/trace_$TAG.txt $P/${TAG}_step_gaps.txt
cp $O/timeline_$TAG.txt $P/${TAG}_step_timeline.txt
cp $O/prof_bench/*/*_kernel_stats.csv $P/${TAG}_bench_graph_kernel_stats.csv
cp $O/prof_bench/*/*_domain_stats.csv $P/${TAG}_bench_graph_domain_stats.csv
cp $O/prof_render/*/*_kernel_stats.csv $P/${TAG}_render_480x270_kernel_stats.csv
cp $O/r06_pmc_traffic.json $P/r06_pmc_traffic.json
ls $P | grep "^${TAG}_\|r06_pmc"
