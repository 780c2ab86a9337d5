mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_attention_r06 -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r6/call26_prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_attention_r06 -name "*kernel_trace.csv" -delete
find $GRAFT_REPO_ROOT/gpurun_out/prof_attention_r06 -name "*kernel_stats.csv" | head -2
