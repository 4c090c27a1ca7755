#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t11.log 2>&1; echo rc=$? >> gpurun_out/r3_t11.log; tail -5 gpurun_out/r3_t11.log
grep -q "rc=0" gpurun_out/r3_t11.log || exit 1
echo "== cfg3"; bash scripts/envab.sh "CRGPU_CAND_FILTER=1 CRGPU_CAND_FILTER=2" --steps 3 --warmup 1 --no-end-to-end
echo "== cfg4"; bash scripts/envab.sh "CRGPU_CAND_FILTER=1 CRGPU_CAND_FILTER=2" --workload cfg4 --steps 3 --warmup 1
timeout -k 10 400 bash scripts/prof.sh r3cf --steps 3 --warmup 1 --no-end-to-end > gpurun_out/r3_cf_prof.log 2>&1; grep -i "group_cand\|k_cp_\|radix_scatter<unsigned int\|radix_hist\|low_support\|umis_tiled" gpurun_out/r3_cf_prof.log | grep calls
