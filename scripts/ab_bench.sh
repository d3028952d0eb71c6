# usage (GPU box): bash scripts/ab_bench.sh [extra bench args]  -> ms/step of 3 runs x 60 steps (graph mode unless overridden)
for i in 1 2 3; do
  python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-stage-timers "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
