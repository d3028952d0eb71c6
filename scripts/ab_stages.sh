# usage (GPU box): bash scripts/ab_stages.sh "stage1 stage2" name1 name2 ... -> per-step totals (us) of the stages and ms/step of the
# standard library and of every soar_amd/_lib/variants/<name>.so, on one box
stages="$1"; shift
run() {
  python "$@" --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us_per_step']; print(' '.join('%s %.1f' % (k, s[k]) for k in '$stages'.split()), ' %.3f ms/step' % d['ms_per_step'])"
}
echo -n "standard      "; run bench.py
for n in "$@"; do echo -n "$n  "; run scripts/ab_lib.py soar_amd/_lib/variants/$n.so; done
echo -n "standard      "; run bench.py
