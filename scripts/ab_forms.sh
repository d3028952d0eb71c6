# usage (GPU box): bash scripts/ab_forms.sh STAGE "ENV1=a ENV2=b" "ENV=c" ...  -> the stage's mean launch duration and ms/step of the
# default build with each set of development switches (and without any), on one box
stage="$1"; shift
run() {
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f us  %.3f ms/step  %s' % (d['roofline']['stage_us']['$stage'], d['ms_per_step'], {k: round(v, 1) for k, v in d['roofline']['stage_us'].items()}))"
}
echo -n "default:  "; run A=1
for e in "$@"; do echo -n "$e:  "; run $e; done
echo -n "default:  "; run A=1
