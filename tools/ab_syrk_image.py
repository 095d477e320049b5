#!/usr/bin/env python3
"""Same-box A/B + correctness of the image accumulation (agpl_syrk.hip) against the float32-staged split kernel and a
float64 reference: python tools/ab_syrk_image.py --n 4000000 --m 512 [--envs "AGPL_LIB_AB=libagpl.so;AGPL_LIB_AB=libagpl_variant.so"] (the library reads no environment variable: variants are builds)."""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4_000_000)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--l", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--gscale", type=float, default=0.25)
ap.add_argument("--envs", default="", help="';'-separated 'VAR=V,VAR2=V2' variants: each runs in a child process")
ap.add_argument("--child", action="store_true")
ap.add_argument("--ref", type=int, default=1)
args = ap.parse_args()

if args.envs and not args.child:
    for v in args.envs.split(";"):
        env = dict(os.environ)
        for kv in v.split(","):
            if kv:
                k, val = kv.split("=")
                env[k] = val
        cmd = [sys.executable, __file__, "--child", "--n", str(args.n), "--m", str(args.m), "--l", str(args.l),
               "--reps", str(args.reps), "--gscale", str(args.gscale), "--ref", str(args.ref)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        print(json.dumps({"variant": v, **(json.loads(line[-1]) if line else {"rc": r.returncode, "err": r.stderr[-800:]})}), flush=True)
    sys.exit(0)

import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ["AGPL_LIB_AB"])

N, M, L = args.n, args.m, args.l
ctx = A.Context(0, seed=1)
lib = _ffi.lib()
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
gam = torch.rand((L, N), device="cuda", generator=g) * args.gscale
bet = torch.randn((L, N), device="cuda", generator=g)
nbytes = lib.agpl_accumulate_image_bytes(C.c_int64(N), C.c_int32(M))
img = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
ctx.call("agpl_accumulate_image", C.c_int64(N), C.c_int32(M), C.c_void_p(Phi.data_ptr()), C.c_void_p(img.data_ptr()))
p = lambda t: C.c_void_p(t.data_ptr())
out = {"N": N, "M": M, "L": L}

def run(image):
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    gg = torch.empty((L, M), dtype=torch.float64, device="cuda")
    call = lambda: ctx.call("agpl_accumulate_split", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Phi),
                            p(img) if image else C.c_void_p(0), p(bet), p(gam), p(G), p(gg))
    call(); call()
    torch.cuda.synchronize()
    lib.agpl_timing_enable(ctx.bind(), 1)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(args.reps):
        call()
    ctx.synchronize()
    out["wall_ms_per_call_" + ("image" if image else "f32staged")] = round((time.perf_counter() - t0) * 1e3 / args.reps, 4)
    tot, cnt = C.c_double(), C.c_int64()
    lib.agpl_timing_read(ctx.bind(), C.c_int32(1), C.byref(tot), C.byref(cnt))
    lib.agpl_timing_enable(ctx.bind(), 0)
    G2, g2 = G.clone(), gg.clone()
    call()
    torch.cuda.synchronize()
    return G2, g2, tot.value / max(cnt.value, 1), bool(torch.equal(G, G2) and torch.equal(gg, g2))

Gi, gi, ms_i, rep_i = run(True)
out["image_ms"] = round(ms_i, 4); out["image_bitwise_repeat"] = rep_i
out["image_symmetric"] = bool(torch.equal(Gi, Gi.transpose(1, 2)))
if args.ref:
    Go, go, ms_o, rep_o = run(False)
    out["f32staged_ms"] = round(ms_o, 4)
    Gr = torch.zeros((L, M, M), dtype=torch.float64, device="cuda"); gr = torch.zeros((L, M), dtype=torch.float64, device="cuda")
    step = 500000
    for l in range(L):
        for i0 in range(0, N, step):
            P = Phi[i0:i0 + step].double()
            Gr[l] += (P * gam[l, i0:i0 + step].double()[:, None]).t() @ P
            gr[l] += P.t() @ bet[l, i0:i0 + step].double()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    if os.environ.get("AGPL_DUMP"):
        E = (Gi[0] - Gr[0]).abs() / Gr[0].abs().max()
        blk = E.reshape(M // 64, 64, M // 64, 64).amax(dim=(1, 3))
        torch.set_printoptions(precision=1, linewidth=220, sci_mode=True)
        print(blk.cpu(), file=sys.stderr)
    out.update(image_relG=rel(Gi, Gr), image_relg=rel(gi, gr), f32staged_relG=rel(Go, Gr), f32staged_relg=rel(go, gr))
print(json.dumps(out), flush=True)
