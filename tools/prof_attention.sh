#!/bin/bash
# kernel stats of the attention-decoder step: tools/prof_attention.sh <tag> -> gpurun_out/prof_att_<tag>/
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_att_$TAG -- python3 $R/tools/bench_attention.py 10 > $R/gpurun_out/prof_att_$TAG.log 2>&1
find $R/gpurun_out/prof_att_$TAG -name "*kernel_trace.csv" -delete
tail -2 $R/gpurun_out/prof_att_$TAG.log
f=$(find $R/gpurun_out/prof_att_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the run")
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
