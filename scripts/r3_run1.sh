#!/bin/bash
# round-3 measurement batch (GPU box): full suite, dense-key suite, cfg3 profile, two A/B runs, the 3M-list bench
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t7.log 2>&1; echo rc=$? >> gpurun_out/r3_t7.log; tail -3 gpurun_out/r3_t7.log
CRGPU_TEST_DENSE=1 python -m pytest tests/test_gpu_count.py tests/test_gpu_configs.py tests/test_gpu_segments.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -m gpu -q 2>&1 | tail -3
bash scripts/prof.sh r3e --steps 3 --warmup 1 > gpurun_out/r3_prof_e.log 2>&1
grep -E "k_group|k_correct_umis|k_giant|k_radix_scatter|k_rl_|k_repair" gpurun_out/r3_prof_e.log | tail -12
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'])" "$1" "$2"; }
show gpurun_out/prof_r3e.json prof
CRGPU_EDGES_MAIN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end --no-default-options > gpurun_out/r3_em.json 2>/dev/null; show gpurun_out/r3_em.json edges_main
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end --no-default-options > gpurun_out/r3_plain2.json 2>/dev/null; show gpurun_out/r3_plain2.json default
python bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > gpurun_out/r3_wl3m2.json 2> gpurun_out/r3_wl3m2.err; show gpurun_out/r3_wl3m2.json wl3m
