#!/usr/bin/env python3
"""Back-to-back launches of each GEMV class: fresh weights every launch (cycling over all
layers, > Infinity Cache) against the same layer's weights every launch (cache resident).
usage: diag_gemv_loop.py [model]      (Q3_LIB selects a diagnostic variant of the library)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3lib as Q
hip = Q.hip_lib()
mdl = sys.argv[1] if len(sys.argv) > 1 else "4B"
os.makedirs("/tmp/q3", exist_ok=True)
path = f"/tmp/q3/{mdl}.bin"
if not os.path.exists(path): Q.synth(mdl, path)
m = hip.q3_model_open(path.encode(), 1024, 0)
hip.forward(m, 1, 0)
L = C.cast(m, C.POINTER(Q.Model)).contents.params.n_layers
print("lib", os.environ.get("Q3_LIB", "product"))
for which in ("qkv", "wo", "gateup", "down"):
    fresh = hip.q3_debug_gemv_loop(m, which.encode(), 0, L, 360)
    same = hip.q3_debug_gemv_loop(m, which.encode(), 3, 4, 360)
    two = hip.q3_debug_gemv_loop(m, which.encode(), 3, 5, 360)
    print(f"{which:7s} fresh {fresh:6.2f} us   same-layer {same:6.2f} us   two-layers {two:6.2f} us", flush=True)
