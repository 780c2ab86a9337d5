for i in 1 2 3 4; do
  # (bench.py no longer reads NSKY_LIB: historical, kept for the record of the round-5 A/B)
  NSKY_LIB=$PWD/scratch/r5/libold.so NSKY_FILM_BWD=8 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-exact-f32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8-wave (old lib)', round(d['ms_per_step'],2))"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-exact-f32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4-wave (tree)    ', round(d['ms_per_step'],2))"
done
