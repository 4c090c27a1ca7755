#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t10.log 2>&1; echo rc=$? >> gpurun_out/r3_t10.log; tail -5 gpurun_out/r3_t10.log
grep -q "rc=0" gpurun_out/r3_t10.log || exit 1
echo "== 737K list"; bash scripts/envab.sh "CRGPU_K2_EAGER_QUAL=1 CRGPU_X=1" --steps 3 --warmup 1 --no-end-to-end
echo "== 3M list"; bash scripts/envab.sh "CRGPU_K2_EAGER_QUAL=1 CRGPU_X=1" --whitelist 6794880 --steps 3 --warmup 1 --no-end-to-end
echo "== cfg2"; bash scripts/envab.sh "CRGPU_K2_EAGER_QUAL=1 CRGPU_X=1" --workload cfg2 --steps 20 --warmup 3
