cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "render_" --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > gpurun_out/pmc1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "render_" --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > gpurun_out/pmc2.log 2>&1
ls gpurun_out/pmc1/*/ gpurun_out/pmc2/*/ | head; tail -2 gpurun_out/pmc1.log
