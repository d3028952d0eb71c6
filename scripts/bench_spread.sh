# GPU box: run-to-run spread of the default bench line -- four processes, each with the host's per-step issue times on stderr
# (SOAR_BENCH_STEP_TIMES=1) and three more timed regions in the same process (SOAR_BENCH_REPEAT); DESIGN.md section 9
for r in 1 2 3 4; do
SOAR_BENCH_STEP_TIMES=1 SOAR_BENCH_REPEAT=3 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2> gpurun_out/spread.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step, host issue %.3f ms/step' % (d['ms_per_step'], d['config']['host_issue_ms_per_step']))"
grep -E "repeat|per step" gpurun_out/spread.err | cut -c1-400
done
