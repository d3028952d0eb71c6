run() { python "$@" --loss avatar --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f us frame_loss per step  %.3f ms/step' % (d['roofline']['stage_us_per_step']['frame_loss'], d['ms_per_step']))"; }
echo -n "standard  "; run bench.py
echo -n "ssimfast  "; run scripts/ab_lib.py soar_amd/_lib/variants/ssimfast.so
echo -n "standard  "; run bench.py
