#!/usr/bin/env python3
"""Phase marks inside one engine launch of the middle layer (diagnostic variant built with
-DQ3_ENG_STAMPS):  make -C qwen3.c_amd variant V=estamps HIPFLAGS_EXTRA=-DQ3_ENG_STAMPS"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["Q3_STAMPS"] = "1"
os.environ.setdefault("Q3_LIB", os.path.join(ROOT, "qwen3.c_amd", "build_estamps", "libq3hip.so"))
import numpy as np
import q3lib as Q
hip = Q.hip_lib()
mdl = sys.argv[1] if len(sys.argv) > 1 else "4B"
path = os.path.join(Q.tmp_dir(), f"{mdl}.bin")
Q.synth(mdl, path)
m = hip.q3_model_open(path.encode(), 1024, 0)
V = m.contents.params.vocab_size
names = ["entry", "Wo: act in regs", "Wo: dot done", "gu: x gathered+normed", "gu: act in regs", "gu: loop done",
         "dn: h quantised", "dn: sync", "dn: dot done", "dn: sync2", "qkv: x gathered+normed", "qkv: done"]
tok = 9707
for pos in range(40):
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, V)
    if pos in (5, 39):
        buf = (C.c_uint64 * 16384)(); hip.q3_debug_stamps(m, buf, 16384)
        w = np.array(buf[:], dtype=np.int64).reshape(256, 64)
        t0 = w[:, 0][w[:, 0] > 0].min()
        print(f"pos {pos}: ns after the earliest consumer entry (min / mean / max over 256 workgroups)")
        for base, who in ((0, "consumer 0"), (16, "consumer 14")):
            print(f" {who}")
            for i, nme in enumerate(names):
                col = w[:, base + i]
                col = col[col > 0]
                if len(col):
                    r = (col - t0) * 10
                    print(f"  {nme:26s} {r.min():7d} {r.mean():9.0f} {r.max():7d}")
        print(" loader: issue time of stream slot s (mean over workgroups), ns")
        ld = w[:, 32:63]
        row = []
        for s in range(31):
            col = ld[:, s]; col = col[col > 0]
            row.append(int(((col - t0) * 10).mean()) if len(col) else -1)
        print("  W :", row[0:4]); print("  G :", row[4:17]); print("  D :", row[17:27]); print("  Q :", row[27:31])
