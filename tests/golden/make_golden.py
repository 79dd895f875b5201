#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REAL reference, compiled here from /root/reference
by oracle/Makefile (gcc -O2 -DNDEBUG, OMP_NUM_THREADS=1 = the deterministic golden build,
SURVEY.md 8(c)).  Only data is written: inputs and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

os.environ["OMP_NUM_THREADS"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import q3lib as Q  # noqa: E402

REF_INC = "/root/reference/include"


def forced_logits(ref, path, feed, seq=0):
    """Teacher-forced token ids `feed` (greedy feedback on a tied random-init model just
    repeats its input token, which would leave the KV cache degenerate)."""
    m = ref.model_create(path.encode(), seq)
    rows = []
    for pos, tok in enumerate(feed):
        rows.append(Q.logits_array(m, ref.forward(m, int(tok), pos)))
    ref.model_free(m)
    return np.stack(rows)


def abi_layout():
    fields = {
        "Q8Tensor": ["s", "q"],
        "ModelParams": ["magic", "version", "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads",
                        "vocab_size", "seq_len", "head_dim", "shared_classifier", "block_size"],
        "ModelWeights": ["wq", "wk", "wv", "wo", "w1", "w2", "w3", "cls", "qe", "fe", "att_rms_norm",
                         "ffn_rms_norm", "out_rms_norm", "q_rms_norm", "k_rms_norm"],
        "ForwardState": ["x", "x_rms_norm", "q", "k", "v", "scores", "mlp_in", "mlp_gate", "logits",
                         "k_cache", "v_cache", "qx", "qh"],
        "Model": ["params", "weights", "state", "data", "size"],
    }
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "model.h"', 'int main(void){']
    for st, fl in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fl:
            src.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "abi.c")
        open(c, "w").write("\n".join(src))
        exe = os.path.join(td, "abi")
        subprocess.check_call(["gcc", "-I" + REF_INC, c, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    return {k: int(v) for k, v in (ln.split() for ln in out.strip().splitlines())}


def main():
    ref = Q.reference_lib()
    assert ref is not None, "needs /root/reference"
    host = Q.host_lib()
    tmp = Q.tmp_dir()
    sums = {}
    for name in ("tiny", "small"):
        path = os.path.join(tmp, f"{name}.bin")
        if os.path.exists(path):
            os.remove(path)
        Q.synth(name, path)
        sums[name] = {"bytes": os.path.getsize(path), "fnv1a64": "%016x" % host.q3_file_checksum(path.encode())}
    json.dump(sums, open(os.path.join(HERE, "checksums.json"), "w"), indent=1)

    frng = np.random.default_rng(7)
    feed = frng.integers(0, 512, size=24).astype(np.int32)
    lg = forced_logits(ref, os.path.join(tmp, "tiny.bin"), feed)
    np.savez_compressed(os.path.join(HERE, "tiny_ref_logits.npz"), feed=feed, logits=lg)
    feed = frng.integers(0, 1024, size=72).astype(np.int32)
    lg = forced_logits(ref, os.path.join(tmp, "small.bin"), feed)
    keep = np.array(list(range(0, 6)) + list(range(62, 72)))
    np.savez_compressed(os.path.join(HERE, "small_ref_logits.npz"), feed=feed, positions=keep, logits=lg[keep])

    # op-level known answers from the reference's own functions
    rng = np.random.default_rng(20250725)
    ops = {}
    n, d = 320, 48
    wq = rng.integers(-127, 128, size=(d, n), dtype=np.int8)
    ws = (0.02 / 73.3 * rng.uniform(0.75, 1.25, size=(d, n // 64))).astype(np.float32)
    x = (rng.standard_normal(n) * 2).astype(np.float32)
    xq = np.zeros(n, np.int8); xs = np.zeros(n // 64, np.float32)
    t = Q.q8view(xq, xs)
    ref.q8_quantize(C.byref(t), Q.fptr(x), n, 64)
    out = np.zeros(d, np.float32)
    wt = Q.q8view(wq.reshape(-1), ws.reshape(-1))
    ref.matmul(Q.fptr(out), C.byref(t), C.byref(wt), n, d, 64)
    ops.update(mm_wq=wq, mm_ws=ws, mm_x=x, mm_xq=xq, mm_xs=xs, mm_out=out)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    o = np.zeros(n, np.float32)
    ref.rmsnorm(Q.fptr(o), Q.fptr(x), Q.fptr(w), n)
    ops.update(rms_w=w, rms_out=o)
    z = (rng.standard_normal(200) * 4).astype(np.float32)
    zs = z.copy(); ref.softmax(Q.fptr(zs), 200)
    ops.update(sm_in=z, sm_out=zs)
    h = rng.standard_normal(128).astype(np.float32)
    for pos in (0, 3, 4097):
        r = h.copy(); ref.rotary(Q.fptr(r), 128, pos)
        ops[f"rope_{pos}"] = r
    ops["rope_in"] = h
    g = (rng.standard_normal(256) * 3).astype(np.float32); u = rng.standard_normal(256).astype(np.float32)
    gs = g.copy(); ref.swiglu(Q.fptr(gs), Q.fptr(u), 256)
    ops.update(sw_g=g, sw_u=u, sw_out=gs)
    np.savez_compressed(os.path.join(HERE, "ops_ref.npz"), **ops)

    json.dump(abi_layout(), open(os.path.join(HERE, "abi_layout.json"), "w"), indent=1, sort_keys=True)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
