for r in 1 2 3 4; do
SOAR_BENCH_STEP_TIMES=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2> gpurun_out/rep.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step host %.3f' % (d['ms_per_step'], d['config']['host_issue_ms_per_step']))"
grep "per step" gpurun_out/rep.err | cut -c1-700
done
