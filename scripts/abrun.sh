#!/bin/bash
# usage (GPU box): bash scripts/abrun.sh "<variant names>" <bench args...>   prints kernel_ms_per_step of each variant, twice
names=$1; shift
for rep in 1 2; do
for v in $names; do
  CRGPU_LIB_PATH=$GRAFT_REPO_ROOT/cellranger_amd/variants/libcrgpu_$v.so python3 bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f'%(k,v) for k,v in d['kernel_ms_per_step'].items()))"
done
done
