#!/usr/bin/env python3
"""Phase marks inside one GEMV launch of the middle layer (diagnostic variant built with
-DQ3_GEMV_STAMPS):  make -C qwen3.c_amd variant V=stamps HIPFLAGS_EXTRA=-DQ3_GEMV_STAMPS
usage: diag_gemv_stamps.py <qkv|wo|gateup|down> [model]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["Q3_STAMPS"] = "1"
os.environ["Q3_STAMP_GEMV"] = sys.argv[1] if len(sys.argv) > 1 else "gateup"
os.environ.setdefault("Q3_LIB", os.path.join(ROOT, "qwen3.c_amd", "build_stamps", "libq3hip.so"))
import q3lib as Q
hip = Q.hip_lib()
mdl = sys.argv[2] if len(sys.argv) > 2 else "4B"
os.makedirs("/tmp/q3", exist_ok=True)
path = f"/tmp/q3/{mdl}.bin"
if not os.path.exists(path): Q.synth(mdl, path)
m = hip.q3_model_open(path.encode(), 1024, 0)
hip.q3_debug_stamps.argtypes = [Q.ModelP, C.POINTER(C.c_uint64), C.c_int]
tok = 9707
names = ["entry", "issued", "x-in|atbar", "prologue|released", "barrier", "scale", "dotted", "end"]
for pos in range(40):
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, 151936)
    if pos in (5, 20, 39):
        buf = (C.c_uint64 * 2048)(); hip.q3_debug_stamps(m, buf, 2048)
        t0 = min(buf[k * 8] for k in range(6) if buf[k * 8])
        print(f"pos {pos}  ({os.environ['Q3_STAMP_GEMV']}; ns after the earliest stamped entry)")
        for k in range(6):
            row = [(buf[k * 8 + i] - t0) * 10 if buf[k * 8 + i] else -1 for i in range(8)]
            print(f"  wg {'first mid last'.split()[k // 2]:5s} wave {'0' if k % 2 == 0 else 'N-1'}: " +
                  "  ".join(f"{n} {v}" for n, v in zip(names, row)))

# every workgroup: entry / barrier open / end, by XCD (workgroup id % 8)
import numpy as np
w = np.array(buf[64:64 + 1024], dtype=np.int64).reshape(256, 4)
w = w[w[:, 2] > 0]
t0 = w[:, 0].min()
e, b, x = (w[:, 0] - t0) * 10, (w[:, 1] - t0) * 10, (w[:, 2] - t0) * 10
print(f"all {len(w)} workgroups: entry {e.min()}..{e.max()}  barrier {b.min()}..{b.max()} (mean {b.mean():.0f})  end {x.min()}..{x.max()} (mean {x.mean():.0f}, p90 {np.percentile(x, 90):.0f})")
for xcd in range(8):
    sel = np.arange(len(w)) % 8 == xcd
    print(f"  xcd {xcd}: entry mean {e[sel].mean():.0f}  barrier mean {b[sel].mean():.0f}  end mean {x[sel].mean():.0f} max {x[sel].max()}")
order = np.argsort(x)
print("  slowest workgroups:", [(int(i), int(x[i])) for i in order[-8:]], " fastest:", [(int(i), int(x[i])) for i in order[:8]])
