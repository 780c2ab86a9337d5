#!/bin/bash
# Same-box A/B of the step time: tools/ab_bench.sh snapshot   (here: copies the tree's package + bench into scratch/base/)
#                                tools/ab_bench.sh            (on the GPU box: bench of scratch/base and of the tree, alternating)
# Box-to-box variation of the step time is ~0.5 ms; two runs on one box agree to ~0.03 ms.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = "snapshot" ]; then
  rm -rf $R/scratch/base && mkdir -p $R/scratch/base/tests
  cp -r $R/neusky_amd $R/bench.py $R/oracle $R/scratch/base/ && cp -r $R/tests/*.py $R/tests/golden $R/scratch/base/tests/
  find $R/scratch/base -name __pycache__ -prune -exec rm -rf {} \;
  echo "snapshot in scratch/base"; exit 0
fi
for i in 1 2 3; do
  for d in $R/scratch/base $R; do
    (cd $d && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-extra-configs 2>/dev/null | tail -1 | python -c "import sys,json; print('$d'.replace('$R','.') or '.', json.loads(sys.stdin.read())['ms_per_step'])")
  done
done
