#!/usr/bin/env python3
"""Sampled decode (q3_generate_sampled: forward + device sampler, no host round trip) next to greedy decode
(q3_generate_greedy) on Qwen3-4B shapes; usage: bench_sampler.py [temperature top_p]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True)
path = "/tmp/q3/4B.bin"
if not os.path.exists(path): Q.synth("4B", path)
m = hip.q3_model_open(path.encode(), 1024, 0)
T = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
P = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
n = 128
out = (C.c_int * n)()
hip.q3_generate_greedy(m, 9707, 0, 16, out)
t0 = time.time(); hip.q3_generate_greedy(m, 9707, 0, n, out); hip.q3_device_sync(m); tg = time.time() - t0
seed = C.c_uint64(1234)
hip.q3_generate_sampled(m, 9707, 0, 16, T, P, C.byref(seed), out)
t0 = time.time(); hip.q3_generate_sampled(m, 9707, 0, n, T, P, C.byref(seed), out); hip.q3_device_sync(m); ts = time.time() - t0
print(f"greedy  {n / tg:8.1f} tok/s  {1e3 * tg / n:.3f} ms/token")
print(f"sampled {n / ts:8.1f} tok/s  {1e3 * ts / n:.3f} ms/token  (temperature {T}, top_p {P}; sampler = {1e3 * (ts - tg) / n:.3f} ms/token on "
      f"random-init logits, whose nucleus spans most of the vocabulary)   distinct tokens {len(set(out))}")
