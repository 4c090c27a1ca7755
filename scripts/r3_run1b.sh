#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_feature_extract.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'])" "$1" "$2"; }
python bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > gpurun_out/r3_wl3m3.json 2> gpurun_out/r3_wl3m3.err; show gpurun_out/r3_wl3m3.json wl3m_dense
python bench.py --whitelist 6794880 --dense-keys 0 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end --no-default-options > gpurun_out/r3_wl3m3n.json 2> gpurun_out/r3_wl3m3n.err; show gpurun_out/r3_wl3m3n.json wl3m_ranks
bash scripts/r3_run2.sh
