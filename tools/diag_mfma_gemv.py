#!/usr/bin/env python3
"""The decode GEMVs on the matrix cores, measured: back-to-back launches (fresh weights every launch, cycling over
all layers) of each GEMV class through k_gemv3 (v_dot4_i32_i8 + DPP, the decode path) and through the prompt path's
int8-MFMA GEMM (v_mfma_i32_16x16x64_i8, K = 64 = one Q8_0 group per instruction; LDS-staged weight slabs) with 1,
16 and 64 activation rows.  Both produce the same bits (tests/test_gpu_prefill.py); at one row 15 of the 16 MFMA
columns carry nothing.  usage: diag_mfma_gemv.py [model]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3lib as Q
hip = Q.hip_lib()
mdl = sys.argv[1] if len(sys.argv) > 1 else "4B"
path = os.path.join(Q.tmp_dir(), f"{mdl}.bin"); Q.synth(mdl, path)
m = hip.q3_model_open(path.encode(), 1024, 0)
hip.forward(m, 1, 0)
p = m.contents.params
L = p.n_layers
P, KVD = p.n_heads * p.head_dim, p.n_kv_heads * p.head_dim
shape = {"qkv": (P + 2 * KVD, p.dim), "wo": (p.dim, P), "gateup": (2 * p.hidden_dim, p.dim), "down": (p.dim, p.hidden_dim)}
print(f"Qwen3-{mdl} shapes, us per launch (HIP events around 360 back-to-back launches), weights fresh from HBM")
print(f"{'matrix':8s} {'bytes':>10s}  {'k_gemv3 (dot4)':>16s}  {'MFMA x1 row':>13s}  {'MFMA x16':>10s}  {'MFMA x64':>10s}")
for which, (dd, nn) in shape.items():
    nbytes = hip.q3_gemv_bytes(dd, nn)
    g = hip.q3_debug_gemv_loop(m, which.encode(), 0, L, 360)
    r = [hip.q3_debug_gemm_loop(m, which.encode(), nt, 0, L, 360) for nt in (1, 16, 64)]
    tb = lambda us: nbytes / us / 1e6
    print(f"{which:8s} {int(nbytes):10d}  {g:8.2f} ({tb(g):4.2f} TB/s)  {r[0]:6.2f} ({tb(r[0]):4.2f})  {r[1]:10.2f}  {r[2]:10.2f}", flush=True)
