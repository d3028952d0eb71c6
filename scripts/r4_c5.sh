export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for w in C5 C2; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w: %.1f frames/s  %.3f ms/step  %s' % (d['value'], d['ms_per_step'], d['config']['plan_form'])); print({k: round(v,1) for k,v in d['roofline']['stage_us_per_step'].items()})"
done
SOAR_PLAN_BATCHED=0 python bench.py --workload C5 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 streams form: %.1f frames/s  %.3f ms/step  %s' % (d['value'], d['ms_per_step'], d['config']['plan_form']))"
