# the graph-replay / eager gradient comparison with the weight gradients on the side stream (and, with an argument, on the main stream too)
mkdir -p gpurun_out/flake
NSKY_ASYNC_WGRAD=1 timeout 600 python tools/flake_graph.py 24 4 > gpurun_out/flake/graph_async1.log 2>&1; echo "async=1 rc=$? $(grep -c MISMATCH gpurun_out/flake/graph_async1.log) mismatches"
if [ -n "$1" ]; then NSKY_ASYNC_WGRAD=0 timeout 600 python tools/flake_graph.py 24 4 > gpurun_out/flake/graph_async0.log 2>&1; echo "async=0 rc=$? $(grep -c MISMATCH gpurun_out/flake/graph_async0.log) mismatches"; fi
grep -h MISMATCH gpurun_out/flake/graph_async*.log | head -12
