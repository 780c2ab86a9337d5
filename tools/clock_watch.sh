#!/bin/bash
# shader clock and package power while the graph-replayed step runs back to back: tools/clock_watch.sh <tag>  ->  gpurun_out/clock_<tag>.txt
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --no-spawn --steps 1500 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-extra-configs > $R/gpurun_out/clock_$TAG.bench 2>&1 &
BP=$!
sleep 20
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power\|busy" | tr '\n' ' '
  echo
  sleep 0.25
done > $R/gpurun_out/clock_$TAG.txt 2>&1
wait $BP
tail -1 $R/gpurun_out/clock_$TAG.bench | cut -c1-300
head -3 $R/gpurun_out/clock_$TAG.txt
