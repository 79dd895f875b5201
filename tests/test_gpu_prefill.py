"""Batched prompt ingestion: the int8-MFMA GEMM against the oracle's matmul row by row,
and q3_prefill against the token-by-token decode path (bit-identical logits and KV cache)."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q
from test_gpu_ops import rand_q8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,ntok", [(64, 16, 1), (128, 48, 3), (320, 64, 16), (2560, 96, 16), (1024, 34, 7),
                                      (9728, 32, 16), (4096, 2560, 5), (2560, 64, 17), (320, 48, 32), (9728, 34, 29), (2560, 64, 64), (1024, 48, 49),
                                      # the LDS-staged kernel: one slab, an odd number of slabs, the 32-row tile (d >= 16384) with ragged rows
                                      (512, 16, 1), (1536, 40, 20), (512, 16400, 33), (1024, 16418, 64)])
def test_mfma_gemm_equals_gemv_per_token(hip, orc, n, d, ntok):
    """asymmetric random data: any slip in the MFMA operand or result maps shows up as a wrong row or token"""
    rng = np.random.default_rng(n + 3 * d + ntok)
    wq, ws = rand_q8(rng, d, n)
    xq, xs = rand_q8(rng, ntok, n, sigma=1.0)
    out = np.zeros((ntok, d), np.float32)
    hip.q3_op_gemm(Q.i8ptr(wq), Q.fptr(ws), Q.i8ptr(xq), Q.fptr(xs), n, d, ntok, Q.fptr(out))
    orc.orc_set_mode(Q.ORC_TREE)
    wt = Q.q8view(wq.reshape(-1), ws.reshape(-1))
    for t in range(ntok):
        xq_t, xs_t = xq[t].copy(), xs[t].copy()           # named: q8view keeps raw pointers
        xt = Q.q8view(xq_t, xs_t)
        tree = np.zeros(d, np.float32)
        orc.orc_matmul(Q.fptr(tree), C.byref(xt), C.byref(wt), n, d, 64)
        assert np.array_equal(out[t], tree), (t,)


@pytest.mark.parametrize("name,n,over", [
    ("tiny", 37, {}), ("small", 50, {}), ("4Bmini", 70, {}),
    # several chunks per position: blocks of positions on one staged K/V tile, in-launch merge (<= 4 chunk slots)
    ("small", 230, {}),
    # more than four chunk slots: the pass merges in the wide second launch
    ("small", 420, {}),
    # eight query heads per kv head (two heads per wave), head_dim 128 and 64
    ("small", 150, {"n_heads": 8}), ("tiny", 200, {"n_heads": 8, "seq_len": 256}),
])
def test_prefill_equals_token_by_token(hip, name, n, over):
    """logits after the prompt, and every later decode step (i.e. the KV cache the prompt left), are
    bit-identical whether the prompt went in 64 positions at a time or one forward() per token"""
    tag = "".join(f"_{k}{v}" for k, v in sorted(over.items()))
    path = os.path.join(Q.tmp_dir(), f"{name}{tag}.bin")
    spec = Q.synth(name, path, **over)
    ma = hip.q3_model_open(path.encode(), 0, 0)
    mb = hip.q3_model_open(path.encode(), 0, 0)
    rng = np.random.default_rng(n)
    V = spec.vocab_size
    n = min(n, spec.seq_len - 8)
    prompt = rng.integers(0, V, size=n).astype(np.int32)
    head = 5                                            # a few decode steps first: the prompt starts at pos 5
    for pos in range(head):
        hip.forward(ma, int(prompt[pos]), pos)
        hip.forward(mb, int(prompt[pos]), pos)
    arr = (C.c_int * (n - head))(*[int(t) for t in prompt[head:]])
    la = Q.logits_array(ma, hip.q3_prefill(ma, arr, n - head, head))
    for pos in range(head, n):
        lb = Q.logits_array(mb, hip.forward(mb, int(prompt[pos]), pos))
    assert np.array_equal(la, lb)
    tok = int(la.argmax())
    for pos in range(n, n + 6):
        la = Q.logits_array(ma, hip.forward(ma, tok, pos))
        lb = Q.logits_array(mb, hip.forward(mb, tok, pos))
        assert np.array_equal(la, lb), pos
        tok = int(la.argmax())
    hip.q3_model_close(ma)
    hip.q3_model_close(mb)
