#!/usr/bin/env python3
"""Development aid: per-loop instruction census of a kernel in a .hip file's gfx950 code (MFMA, LDS reads, DMA pieces, barriers,
scratch traffic, vmcnt waits): python tools/isa_loops.py <file.hip> <kernel name substring> [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, kname, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
mk = open(os.path.join(ROOT, "augmentedgplikelihoods.jl_amd", "csrc", "Makefile")).read()
flags = re.search(r"^COMMON\s*:=\s*(.*)$", mk, flags=re.M).group(1).replace("$(ARCH)", "gfx950").split()
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    asm = open(out).read().splitlines()
start = next(i for i, l in enumerate(asm) if re.match(r"^_Z\w*" + kname + r"\w*:", l))
end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
body = [l.split(";")[0].rstrip() for l in asm[start:end + 1]]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
print(len(body), "lines; scratch instructions in the kernel:", sum("scratch_" in x for x in body))
seen = set()
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = body[a:i + 1]
        nm = sum("v_mfma" in x for x in seg)
        if nm >= 10 and a not in seen:
            seen.add(a)
            waits = sorted(set(re.findall(r"vmcnt\((\d+)\)", " ".join(seg))))
            ngl = sum(x.strip().startswith("global_load_dword") for x in seg)
            print(f"loop {a}-{i}: mfma {nm} ds_read {sum('ds_read' in x for x in seg)} dma {sum('global_load_lds' in x for x in seg)} "
                  f"gload {ngl} barrier {sum('s_barrier' in x for x in seg)} "
                  f"scratch {sum('scratch_' in x for x in seg)} vmcnt {waits} lines {i - a}")
