cd $GRAFT_REPO_ROOT
for v in 0 4; do
  export CRGPU_ABLATE=$v
  rm -rf gpurun_out/prof_abl$v
  bash scripts/prof.sh abl$v --workload cfg3 --reads-per-gpu 200000000 --steps 2 --warmup 1 2>&1 | grep -E "^k_correct_umis" | sed "s/^/ablate=$v  /"
done
