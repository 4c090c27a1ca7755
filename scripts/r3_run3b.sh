#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_count.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'])" "$1" "$2"; }
python bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > gpurun_out/r3_wl3m4.json 2> gpurun_out/r3_wl3m4.err; show gpurun_out/r3_wl3m4.json wl3m_low10
bash scripts/r3_run3.sh
