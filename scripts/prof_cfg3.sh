cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python3 bench.py --workload cfg3 --reads-per-gpu 200000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_cfg3.json 2> gpurun_out/prof_cfg3.err
ls -R gpurun_out/prof_cfg3 | head -20
