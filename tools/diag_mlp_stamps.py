#!/usr/bin/env python3
"""Phase marks inside one k_mlp launch of the middle layer (diagnostic variant built with
-DQ3_MLP_STAMPS):  make -C qwen3.c_amd variant V=mstamps HIPFLAGS_EXTRA=-DQ3_MLP_STAMPS"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["Q3_STAMPS"] = "1"
os.environ["Q3_STAMP_GEMV"] = "mlp"
os.environ.setdefault("Q3_LIB", os.path.join(ROOT, "qwen3.c_amd", "build_mstamps", "libq3hip.so"))
import numpy as np
import q3lib as Q
hip = Q.hip_lib()
mdl = sys.argv[1] if len(sys.argv) > 1 else "4B"
path = os.path.join(Q.tmp_dir(), f"{mdl}.bin")
Q.synth(mdl, path)
m = hip.q3_model_open(path.encode(), 1024, 0)
V = m.contents.params.vocab_size
names = ["entry", "prologue done (w0)", "gate/up loop end (wN)", "barrier3 (w0)", "h stored+drained (w0)", "grid wait done (w0)",
         "barrier4 (wN)", "h fetched+quantised (wN)", "barrier5 (wN)", "end (wN)"]
tok = 9707
for pos in range(40):
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, V)
    if pos in (5, 20, 39):
        buf = (C.c_uint64 * 4096)(); hip.q3_debug_stamps(m, buf, 4096)
        w = np.array(buf[:], dtype=np.int64).reshape(256, 16)[:, :10]
        t0 = w[:, 0].min()
        r = (w - t0) * 10
        print(f"pos {pos}: ns after the earliest workgroup entry (min / mean / max over 256 workgroups)")
        for i, nme in enumerate(names):
            print(f"  {nme:28s} {r[:, i].min():7d} {r[:, i].mean():9.0f} {r[:, i].max():7d}")
        order = np.argsort(r[:, 4])
        print("   last to publish:", [(int(i), int(r[i, 4])) for i in order[-6:]], " first:", [(int(i), int(r[i, 4])) for i in order[:4]])
