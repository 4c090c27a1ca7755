#!/bin/bash
# round-3 official lines (GPU box): bash scripts/r3_official.sh.  Every step under its own timeout; progress goes to stdout as it happens.
cd $GRAFT_REPO_ROOT
T="timeout -k 10 420"
show() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[2],d['ms_per_step'],d['value'],d.get('default_options_ms_per_step'),d['kernel_ms_per_step'])" "$1" "$2"; }
$T python bench.py --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r03_cfg4_500M_bench.json 2> gpurun_out/r03_cfg4.err && show gpurun_out/r03_cfg4_500M_bench.json cfg4 || { echo "cfg4 failed"; tail -3 gpurun_out/r03_cfg4.err; }
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03cfg4 -- python3 bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-default-options > gpurun_out/r03_cfg4_prof.json 2> gpurun_out/r03_cfg4_prof.err && show gpurun_out/r03_cfg4_prof.json cfg4_prof
$T python bench.py --steps 5 --warmup 2 > gpurun_out/r03_cfg3_1B_bench.json 2> gpurun_out/r03_cfg3.err && show gpurun_out/r03_cfg3_1B_bench.json cfg3 || { echo "cfg3 failed"; tail -3 gpurun_out/r03_cfg3.err; }
$T bash scripts/prof.sh r03 --steps 3 --warmup 1 --no-end-to-end > gpurun_out/r03_prof.log 2>&1; show gpurun_out/prof_r03.json cfg3_prof
$T python bench.py --workload cfg2 --steps 20 --warmup 3 > gpurun_out/r03_cfg2_100M_bench.json 2> gpurun_out/r03_cfg2.err && show gpurun_out/r03_cfg2_100M_bench.json cfg2
$T python bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > gpurun_out/r03_cfg3_1B_wl3m_bench.json 2> gpurun_out/r03_wl3m.err && show gpurun_out/r03_cfg3_1B_wl3m_bench.json wl3m
$T python bench.py --dupinfo --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_cfg3_1B_dupinfo_bench.json 2> gpurun_out/r03_dup.err && show gpurun_out/r03_cfg3_1B_dupinfo_bench.json dupinfo
$T python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_cfg5_500M_bench.json 2> gpurun_out/r03_cfg5.err && show gpurun_out/r03_cfg5_500M_bench.json cfg5
$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03wl3m -- python3 bench.py --whitelist 6794880 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end --no-default-options > gpurun_out/r03_wl3m_prof.json 2> gpurun_out/r03_wl3m_prof.err && show gpurun_out/r03_wl3m_prof.json wl3m_prof
timeout -k 10 900 bash scripts/pmc_families.sh r03 --steps 1 --warmup 1 > gpurun_out/r03_pmc.log 2>&1; tail -12 gpurun_out/r03_pmc.log
