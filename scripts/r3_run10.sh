#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t13.log 2>&1; echo rc=$? >> gpurun_out/r3_t13.log; tail -5 gpurun_out/r3_t13.log
grep -q "rc=0" gpurun_out/r3_t13.log || exit 1
echo "== cfg4"; bash scripts/envab.sh "CRGPU_X=1" --workload cfg4 --steps 3 --warmup 1
echo "== cfg3"; bash scripts/envab.sh "CRGPU_X=1" --steps 3 --warmup 1 --no-end-to-end
