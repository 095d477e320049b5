#!/usr/bin/env python3
"""Sustained float16 MFMA rate of the device (agpl_probe_mfma, float16 operands), one JSON line per (mode, workgroups per CU).
Usage: python3 tools/probe_mfma.py [iters]"""
import ctypes as C, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agpl_amd as A
from agpl_amd import _ffi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = A.Context(seed=1)
for mode in (0, 1, 2, 3):
    for wpc in ((1, 2, 4) if mode < 2 else (1, 2)):  # (the 16x16x32 probe holds 2 workgroups per CU)
        tf, ms = C.c_double(0), C.c_double(0)
        _ffi.check(ctx.bind(), _ffi.lib().agpl_probe_mfma(ctx.bind(), C.c_int32(_ffi.F32), C.c_int32(iters // wpc // (1 if mode < 2 else 4)), C.c_int32(mode),
                                                              C.c_int32(wpc), C.byref(tf), C.byref(ms)))
        print(json.dumps({"mode": ["32x32x16", "32x32x16 + lds fragment reads", "16x16x32", "16x16x32 + lds fragment reads"][mode], "workgroups_per_cu": wpc,
                          "waves_per_simd": wpc, "tflops": round(tf.value, 1), "ms_per_launch": round(ms.value, 3),
                          "frac_of_2500": round(tf.value / 2500, 3)}), flush=True)
