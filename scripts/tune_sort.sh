cd $GRAFT_REPO_ROOT
for it in 8 12 16 24; do
  export CRGPU_EXTRA_FLAGS="-DSORT_ITEMS=$it"
  python -m cellranger_amd.build --force > /dev/null 2>&1
  rm -rf gpurun_out/prof_tune$it
  bash scripts/prof.sh tune$it --workload cfg3 --reads-per-gpu 200000000 --steps 2 --warmup 1 2>&1 | grep -E "k_radix_scatter<unsigned (long|int)" | grep calls | sed "s/^/items=$it  /"
done
