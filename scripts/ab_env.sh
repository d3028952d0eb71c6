# GPU box: the default bench line with an environment switch off / on, twice each, interleaved.  usage: bash scripts/ab_env.sh NAME [bench.py arguments]
name=$1; shift
run() { python bench.py "$@" --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %.3f ms/step  %.1f frames/s' % (d['ms_per_step'], d['value']))"; }
for r in 1 2; do
  echo -n "$name=0 "; env $name=0 bash -c "$(declare -f run); run $*"
  echo -n "$name=1 "; env $name=1 bash -c "$(declare -f run); run $*"
done
