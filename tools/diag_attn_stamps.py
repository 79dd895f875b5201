"""Device-clock marks inside the attention kernel of the last layer, every workgroup
(variant build: make -C qwen3.c_amd variant V=astamps HIPFLAGS_EXTRA=-DQ3_ATTN_STAMPS).
Marks: 0 entry, 1 K tile requested + pos known, 2 head norms done, 3 K tile in LDS,
4 scores + softmax done, 5 V tile in LDS, 6 PV done, 7 end (partials published / merged).
CTX=n sets the number of cached positions."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["Q3_STAMPS"] = "1"
os.environ["Q3_GRAPH"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ.setdefault("Q3_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qwen3.c_amd", "build_astamps", "libq3hip.so"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True); path = "/tmp/q3/4B.bin"; Q.synth("4B", path)
CTX = int(os.environ.get('CTX', '0'))
m = hip.q3_model_open(path.encode(), max(1024, CTX + 128), 0)
if CTX: hip.q3_kv_fill_random(m, CTX, 5)
hip.q3_debug_stamps.argtypes = [Q.ModelP, C.POINTER(C.c_uint64), C.c_int]
NWG = 8 * 64
tok = 9707
for pos in range(CTX, CTX + 40):
    buf = (C.c_uint64 * (8 * NWG))()
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, 151936)
    if pos - CTX in (0, 5, 20, 39):
        hip.q3_debug_stamps(m, buf, 8 * NWG)
        a = np.array(buf[:], dtype=np.int64).reshape(NWG, 8)
        nch = pos // 64 + 1
        live = np.array([w for w in range(NWG) if (w // 8) < nch and a[w, 7] >= a[w, 0] > 0])
        t0 = a[live, 0].min()
        r = (a[live] - t0) * 10          # ns (100 MHz clock)
        print(f"pos {pos}: {len(live)} workgroups; ns after the first entry")
        for i in range(8):
            print(f"  mark {i}: min {r[:, i].min():6d}  mean {r[:, i].mean():8.0f}  max {r[:, i].max():6d}")
        print("  workgroup (0,0):", list(r[0]))
        dur = r[:, 7] - r[:, 0]
        print(f"  per-workgroup duration: min {dur.min()} mean {dur.mean():.0f} max {dur.max()}")
