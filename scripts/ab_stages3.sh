# usage (GPU box): bash scripts/ab_stages3.sh name1 name2 ... -> stage table of the standard library and of soar_amd/_lib/variants/<name>.so, twice each, interleaved
run() {
  python "$@" --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us_per_step']; print(' %.3f ms/step  %.1f frames/s ' % (d['ms_per_step'], d['value']), ' '.join('%s %.1f' % (k.replace('render_','').replace('lbs_','')[:9], v) for k, v in s.items()))"
}
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for rep in 1 2; do
echo -n "standard   "; run bench.py
for n in "$@"; do echo -n "$n  "; run scripts/ab_lib.py soar_amd/_lib/variants/$n.so; done
done
