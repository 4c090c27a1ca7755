#!/bin/bash
# usage (GPU box): bash scripts/envab.sh "<VAR=a> <VAR=b> ..." <bench args...> : same library, different environment, twice each
envs=$1; shift
for rep in 1 2; do
for ev in $envs; do
  env $ev timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ev', 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f'%(k,v) for k,v in d['kernel_ms_per_step'].items()))"
done
done
