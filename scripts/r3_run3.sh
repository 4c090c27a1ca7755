#!/bin/bash
# round-3 measurement batch 3 (GPU box): PMC passes of cfg3 (keys-only and --dupinfo), three --dupinfo runs, rank rehearsal
cd $GRAFT_REPO_ROOT
bash scripts/pmc_families.sh r3 --workload cfg3 --steps 1 --warmup 1 > gpurun_out/r3_pmc.log 2>&1; tail -12 gpurun_out/r3_pmc.log
bash scripts/pmc_families.sh r3dup --workload cfg3 --dupinfo --steps 1 --warmup 1 > gpurun_out/r3_pmc_dup.log 2>&1; tail -12 gpurun_out/r3_pmc_dup.log
for k in 1 2 3; do python bench.py --dupinfo --steps 5 --warmup 2 --no-cpu-baseline --no-default-options > gpurun_out/r3_dupinfo_$k.json 2>/dev/null; python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print('dupinfo',d['ms_per_step'],d['host_step_marks_ms'],d['kernel_ms_per_step'])" gpurun_out/r3_dupinfo_$k.json; done
python3 scripts/rehearse_ranks.py 4 100000000 > gpurun_out/r3_ranks4.log 2>&1; tail -5 gpurun_out/r3_ranks4.log
