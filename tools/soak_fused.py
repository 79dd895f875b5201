#!/usr/bin/env python3
"""Long soak of the in-launch hand-offs: N greedy tokens on full-size 4B shapes from position 0 (one-chunk shapes,
in-launch merge, loader-wave attention + merge/Wo launch, more than one chunk per workgroup past 4096 positions) with
the fused launches, against the same run with separate attention and Wo launches (Q3_FUSE=0): the token streams must
be identical -- one stale granule anywhere in 36 x N hand-offs x 248 consumers would fork them.
usage: soak_fused.py [N=6000] [model=4B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
mdl = sys.argv[2] if len(sys.argv) > 2 else "4B"
path = os.path.join(Q.tmp_dir(), f"{mdl}.bin"); Q.synth(mdl, path)
def run(fuse):
    if not fuse: os.environ["Q3_FUSE"] = "0"
    try:
        m = hip.q3_model_open(path.encode(), N + 8, 0)
        assert hip.q3_device_attach(m) == 0
    finally:
        os.environ.pop("Q3_FUSE", None)
    out = (C.c_int * N)()
    got = 0
    while got < N:                       # blocks of 1000 tokens: progress lines for the watchdog
        n = min(1000, N - got)
        blk = (C.c_int * n)()
        tok = 9707 if got == 0 else out[got - 1]
        assert hip.q3_generate_greedy(m, tok, got, n, blk) == n
        for i in range(n): out[got + i] = blk[i]
        got += n
        print(f"  {'fused' if fuse else 'separate'}: {got} tokens", flush=True)
    hip.q3_model_close(m)
    return np.array(out[:], dtype=np.int64)
a, b = run(True), run(False)
same = bool(np.array_equal(a, b))
first = int(np.argmax(a != b)) if not same else -1
print(f"{mdl}: {N} greedy tokens from position 0, fused vs separate launches: {'IDENTICAL' if same else f'FORK at token {first}'}; "
      f"{len(set(a.tolist()))} distinct tokens in the stream", flush=True)
sys.exit(0 if same else 1)
