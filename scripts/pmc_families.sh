#!/bin/bash
# usage (GPU box): bash scripts/pmc_families.sh <tag> <bench args...>
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section; no other trace domain
# next to --pmc) -> gpurun_out/pmc_<tag>_fetch_write.json: per-kernel averages and, per kernel family of bench.py's
# ledger, the HBM bytes per element: read bytes = 2 * FETCH_SIZE * 1024 on gfx950, write bytes = WRITE_SIZE * 1024.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  timeout -k 5 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py "$@" --no-cpu-baseline --no-verify --no-end-to-end --no-default-options > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
done
python3 - "$tag" "$*" <<'PY'
import csv, glob, collections, json, sys
tag, args = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, c)):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                res[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
bench = json.loads(open("gpurun_out/pmc_%s_FETCH_SIZE.json" % tag).read().strip().splitlines()[-1])
def family(k):
    if k.startswith("void k_radix_scatter<unsigned long") : return "sort_scatter"
    if "k_build_keys" in k: return "keys"
    if any(x in k for x in ("k_lookup_hot", "k_stage_idx", "k_hist_buckets", "k_match_binned", "k_hot_", "k_match<")): return "match"
    if any(x in k for x in ("k_correct_records", "k_collect_miss", "k_correct<", "k_region_offsets", "k_correct_sorted", "k_compact_records",
                            "k_flag_corrected")): return "correct"
    if any(x in k for x in ("k_csc", "SeenFlag")): return "matrix"
    if any(x in k for x in ("k_extract_", "k_feature_counts", "k_match_features")): return "feature"
    if any(x in k for x in ("k_find_descents", "k_repair_runs", "k_order_runs", "k_global_hist")): return "sort_hist"
    if any(x in k for x in ("k_cp_", "k_correct_umis", "k_giant", "k_rep_", "k_group_", "k_low_support", "k_triplets", "k_radix_scatter<unsigned int",
                            "k_radix_hist<unsigned int", "k_per_read", "k_unpack", "k_corrected_reads", "k_mt_", "k_rl_", "k_trip_counts")): return "dedup"
    return None
kern, fam = [], collections.defaultdict(float)
for k, v in sorted(res.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0])) - sum(kv[1].get("WRITE_SIZE", [0]))):
    f, w = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
    kern.append({"kernel": k, "family": family(k), "launches": len(f), "fetch_size_kb_avg": sum(f) / max(len(f), 1),
                 "write_size_kb_avg": sum(w) / max(len(w), 1),
                 "hbm_bytes_total": (2 * sum(f) + sum(w)) * 1024})
    if family(k): fam[family(k)] += (2 * sum(f) + sum(w)) * 1024
n_steps = bench["steps"] + bench["warmup"]
units = bench.get("kernel_units_per_step", {})
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py %s "
                 "(scripts/pmc_families.sh)" % args,
       "note": "units KB as reported; on gfx950 FETCH_SIZE counts half of a coalesced streaming read (MI355X_MICROARCH.md, "
               "HBM): hbm_read_bytes = 2*FETCH_SIZE*1024, hbm_write_bytes = WRITE_SIZE*1024.  Per family: bytes of all its "
               "launches / (steps + warm-up steps) / elements per step of bench.py's ledger",
       "steps_profiled": n_steps, "kernels": kern, "derived": {}}
for name, b in fam.items():
    out["derived"][name + "_hbm_bytes_per_step"] = b / n_steps
    if units.get(name):
        out["derived"][name + "_hbm_bytes_per_element"] = b / n_steps / units[name]
json.dump(out, open("gpurun_out/pmc_%s_fetch_write.json" % tag, "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
PY
