#!/usr/bin/env python3
"""profiles/rNN_pmc_sq_*.json from a rocprofv3 --pmc pass with SQ / GRBM counters (see profiles/README.md).
Usage: pmc_sq_json.py <pmc_dir> <out.json> [note]
Derived per kernel: effective clock = GRBM_GUI_ACTIVE / 8 / duration (the counter is summed over the 8 XCDs), MFMA pipe
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), and the split of wave-cycles into
waiting (s_waitcnt / barrier), issue-stalled and issuing (SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY)."""
import collections, csv, glob, json, sys

root, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
KEEP = ("marginal_factor_queue_kernel", "marginal_factor_persist_kernel", "marginal_split256_kernel", "marginal_factor16_kernel", "syrk_strip_kernel",
        "syrk_split_kernel", "factor_pipe_kernel", "factor_kernel", "aux_sample_pg1_retry_kernel", "reduce_slab_kernel", "aux_sample_pg1_kernel", "aux_sample_kernel", "gibbs_project_image_kernel", "gibbs_project_kernel",
        "gibbs_sample_kernel", "agpl_fused_point_kernel")


def short(n):
    for k in KEEP:
        if k in n:
            return k
    return None


for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
res = {"note": note, "kernels": {}}
for k, d in acc.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    t = sum(dur[k]) / max(len(dur[k]), 1)
    e = {"avg_ms": round(t * 1e3, 4), "launches": len(dur[k]), "counters": c}
    if "GRBM_GUI_ACTIVE" in c and t > 0:
        e["effective_clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_pipe_utilisation"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n in c:
                e["frac_" + n] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
    # in-flight-count accumulators / instruction counts: mean latency in the counter's own tick (the ratio of the two
    # is unit-free: how many LDS round trips one global load takes)
    if c.get("SQ_INSTS_VMEM_RD") and "SQ_INST_LEVEL_VMEM" in c:
        e["vmem_latency_ticks"] = round(c["SQ_INST_LEVEL_VMEM"] / c["SQ_INSTS_VMEM_RD"], 2)
    if c.get("SQ_INSTS_LDS") and "SQ_INST_LEVEL_LDS" in c:
        e["lds_latency_ticks"] = round(c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"], 2)
        if "vmem_latency_ticks" in e:
            e["vmem_over_lds_latency"] = round(e["vmem_latency_ticks"] / e["lds_latency_ticks"], 1)
    # vector-ALU occupancy of the SIMDs: SQ_ACTIVE_INST_VALU counts (per SIMD) the cycles a VALU instruction is executing
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if c.get("SQ_INSTS_VALU") and "SQ_WAVE_CYCLES" in c:
        e["valu_insts_per_wave_cycle"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
        e["lds_busy_frac_of_cu_cycles"] = round(c["SQ_LDS_IDX_ACTIVE"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256), 3)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in res["kernels"].items()}, indent=1))
