#!/bin/bash
# round-3 measurement batch 2 (GPU box): cfg4 at full size with kernel stats, cfg5, feature-extraction throughput + kernel stats
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'],d.get('output'))" "$1" "$2"; }
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3cfg4 -- python3 bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-default-options > gpurun_out/r3_cfg4_500M_prof.json 2> gpurun_out/r3_cfg4_500M_prof.err
show gpurun_out/r3_cfg4_500M_prof.json cfg4_prof; tail -2 gpurun_out/r3_cfg4_500M_prof.err
python bench.py --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r3_cfg4_500M.json 2> gpurun_out/r3_cfg4_500M.err; show gpurun_out/r3_cfg4_500M.json cfg4; tail -2 gpurun_out/r3_cfg4_500M.err
python bench.py --workload cfg5 --steps 5 --warmup 2 > gpurun_out/r3_cfg5_500M.json 2> gpurun_out/r3_cfg5_500M.err; show gpurun_out/r3_cfg5_500M.json cfg5; tail -2 gpurun_out/r3_cfg5_500M.err
python scripts/fx_bench.py 8000000 96 > gpurun_out/r3_fx_final.log 2>&1; cat gpurun_out/r3_fx_final.log
python scripts/fx_bench.py 8000000 32 > gpurun_out/r3_fx_final32.log 2>&1; cat gpurun_out/r3_fx_final32.log
