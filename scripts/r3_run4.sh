#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo rc=$? >> gpurun_out/r3_t8.log; tail -3 gpurun_out/r3_t8.log
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'])" "$1" "$2"; }
python bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > gpurun_out/r3_wl3m5.json 2> gpurun_out/r3_wl3m5.err; show gpurun_out/r3_wl3m5.json wl3m_low10
python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_cfg4_resume.json 2> gpurun_out/r3_cfg4_resume.err; show gpurun_out/r3_cfg4_resume.json cfg4_resume; tail -2 gpurun_out/r3_cfg4_resume.err
