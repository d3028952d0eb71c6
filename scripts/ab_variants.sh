# usage (GPU box): bash scripts/ab_variants.sh STAGE name1 name2 ...  -> the stage's mean launch duration (us) and ms/step of the
# standard library and of every soar_amd/_lib/variants/<name>.so, on one box
stage="$1"; shift
run() {
  python "$@" --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f us  %.3f ms/step' % (d['roofline']['stage_us']['$stage'], d['ms_per_step']))"
}
echo -n "standard      "; run bench.py
for n in "$@"; do echo -n "$n  "; run scripts/ab_lib.py soar_amd/_lib/variants/$n.so; done
echo -n "standard      "; run bench.py
