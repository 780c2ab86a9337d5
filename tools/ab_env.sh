#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_env.sh VAR val1 val2 [repeats]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-exact-f32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), round(d['ms_per_step'],2))"
  done
done
