#!/usr/bin/env python3
"""Prompt ingestion rate of q3_prefill (up to 64 positions per pass, int8 MFMA) next to feeding the same
prompt through forward() token by token, Qwen3-4B shapes.  usage: bench_prefill.py [prompt_len]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True)
path = "/tmp/q3/4B.bin"
if not os.path.exists(path): Q.synth("4B", path)
m = hip.q3_model_open(path.encode(), 2048, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prompt = np.random.default_rng(0).integers(0, 151936, size=n).astype(np.int32)
arr = (C.c_int * n)(*[int(t) for t in prompt])
hip.q3_prefill(m, arr, min(n, 32), 0)                  # warm-up
t0 = time.time(); lg = hip.q3_prefill(m, arr, n, 0); tp = time.time() - t0
a = Q.logits_array(m, lg)
for pos in range(8): hip.forward(m, int(prompt[pos]), pos)
t0 = time.time()
for pos in range(n): lg = hip.forward(m, int(prompt[pos]), pos)
tf = time.time() - t0
b = Q.logits_array(m, lg)
print(f"prefill {n} tokens: {n / tp:8.1f} tok/s ({1e3 * tp:.1f} ms)   token by token: {n / tf:8.1f} tok/s ({1e3 * tf:.1f} ms)   "
      f"x{tf / tp:.2f}   same logits: {np.array_equal(a, b)}")
