#!/usr/bin/env python3
"""Soak: graph replay vs eager launches of the same model over thousands of positions (all three attention
launch shapes), and a long prompt through q3_prefill vs token by token -- every logit bit-identical."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True)
path = "/tmp/q3/soak.bin"
Q.synth("4Bmini", path, seq_len=4096, vocab_size=4096)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
a = hip.q3_model_open(path.encode(), 0, 0)       # graph replay
b = hip.q3_model_open(path.encode(), 0, 0)       # eager (layer taps force plain launches)
hip.q3_tap_enable(b, 1)
tok, bad = 11, 0
for pos in range(N):
    la = Q.logits_array(a, hip.forward(a, tok, pos))
    lb = Q.logits_array(b, hip.forward(b, tok, pos))
    if not np.array_equal(la, lb):
        bad += 1
        print("MISMATCH at pos", pos, flush=True)
        if bad > 3: break
    tok = int(la.argmax()) if pos % 7 else int((pos * 2654435761) % 4096)
print(f"graph vs eager: {N} positions, mismatches {bad}", flush=True)
c = hip.q3_model_open(path.encode(), 0, 0)
prompt = np.random.default_rng(1).integers(0, 4096, size=2500).astype(np.int32)
arr = (C.c_int * len(prompt))(*[int(t) for t in prompt])
lc = Q.logits_array(c, hip.q3_prefill(c, arr, len(prompt), 0))
d = hip.q3_model_open(path.encode(), 0, 0)
for pos, t in enumerate(prompt):
    ld = hip.forward(d, int(t), pos)
ld = Q.logits_array(d, ld)
print("prefill 2500 == token by token:", np.array_equal(lc, ld), flush=True)
sys.exit(1 if bad or not np.array_equal(lc, ld) else 0)
