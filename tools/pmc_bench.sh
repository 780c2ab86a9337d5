#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass),
# counters only with --kernel-trace, eager launches (one dispatch record per kernel launch).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$c -- python3 $R/bench.py --no-spawn --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-exact-f32 --no-extra-configs --no-live-pmc > $R/gpurun_out/pmc_bench_$c.log 2>&1
done
ls $R/gpurun_out/pmc_bench_FETCH_SIZE/*
