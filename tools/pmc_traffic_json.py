#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic_*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; see profiles/README.md).
Usage: pmc_traffic_json.py <fetch_dir> <write_dir> <kernel_stats.csv> <out.json> [N M [lik L]]   (default: C2 = 10000000 512 bernoulli 1)"""
import collections, csv, glob, json, re, sys

def means(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

KNOWN = ("marginal_factor_queue_kernel", "marginal_factor_persist_kernel", "marginal_split256_kernel", "syrk_gang_kernel", "syrk_strip_kernel", "syrk_split_kernel",
         "agpl_fused_point_kernel", "reduce_slab_kernel", "reduce_G_kernel", "gibbs_project_image_kernel", "gibbs_project_kernel", "gibbs_sample_kernel", "factor_pipe_kernel", "factor_kernel", "aux_sample_pg1_retry_kernel",
         "split_prep_kernel", "acc_prep_kernel", "aux_sample_pg1_kernel", "aux_sample_kernel")


def short(n):
    for k in KNOWN:  # (template instantiations arrive mangled)
        if k in n:
            return k
    return n[:40]

fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
avg = {}
for r in csv.DictReader(open(sys.argv[3])):
    avg.setdefault(short(r["Name"]), float(r["AverageNs"]) / 1e6)
N, M = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (10_000_000, 512)
LIK, L = (sys.argv[7], int(sys.argv[8])) if len(sys.argv) > 8 else ("bernoulli", 1)
alg = {"marginal_split256_kernel": N * M * 4, "syrk_split_kernel": N * M * 4,
       "marginal_factor_persist_kernel": N * M * 4, "marginal_factor_queue_kernel": N * M * 4, "syrk_strip_kernel": N * M * 4,
       "syrk_gang_kernel": N * M * 4, "gibbs_project_kernel": N * (M * 4 + 8), "gibbs_project_image_kernel": N * (M * 4 + 8),
       "gibbs_sample_kernel": N * 24, "reduce_slab_kernel": None, "agpl_fused_point_kernel": N * (2 * 2 * 4 + 4 + 1 + 8),
       "aux_sample_pg1_kernel": N * 16}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 --no-cpu "
               f"--no-parity ({LIK} N={N} M={M}). FETCH_SIZE is reported in KiB and doubled per "
               "MI355X_MICROARCH.md 'HBM' (16 B/lane coalesced reads report 1/2; calibrated on reduce_slab_kernel: "
               "1.60 GB of slabs read); WRITE_SIZE (KiB) is taken as is.",
       "config": {"lik": LIK, "N": N, "M": M, "L": L}, "kernels": {}}
for k in KNOWN:
    if k not in fetch:
        continue
    fb, wb = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_size_kib_raw": fetch[k], "fetch_bytes_corrected": fb, "write_bytes": wb,
                         "algorithmic_bytes": alg.get(k), "avg_ms": round(avg.get(k, float("nan")), 4),
                         "traffic_bytes": fb + wb}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
