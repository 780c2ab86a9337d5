#!/bin/bash
# SQ counters of the chain kernels (matrix-pipe busy cycles, vector/matrix co-execution, wait buckets): tools/pmc_sq.sh <tag> [program args...]
# one pass per counter set (8 SQ slots), counters only with --kernel-trace.  Output: gpurun_out/sq_<tag>_<n>/ + gpurun_out/sq_<tag>.txt
TAG=${1:-x}; shift
PROG=${@:-bench.py --no-spawn --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-exact-f32 --no-extra-configs}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/gpurun_out/sq_counters_avail.txt
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC")
n=0
for s in "${SETS[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --output-format csv -d $R/gpurun_out/sq_${TAG}_$n -- python3 $R/$PROG > $R/gpurun_out/sq_${TAG}_$n.log 2>&1
  n=$((n+1))
done
python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sq_${TAG}_0 $R/gpurun_out/sq_${TAG}_1 > $R/gpurun_out/sq_$TAG.txt 2>&1
find $R/gpurun_out/sq_${TAG}_0 $R/gpurun_out/sq_${TAG}_1 -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/sq_$TAG.txt
