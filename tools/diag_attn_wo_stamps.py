"""Device-clock marks inside the fused attention + Wo launch (k_attn_wo) of the last layer: the attention
workgroups' eight marks (tools/diag_attn_stamps.py) and, for every consumer workgroup, 0 entry, 1 Wo rows
requested (after the hold-back), 2 L2 warm-up issued, 3 first granules tagged, 4 all granules in, 5 workgroup
met, 6 dots + residual stored.  Variant build: make -C qwen3.c_amd variant V=astamps HIPFLAGS_EXTRA=-DQ3_ATTN_STAMPS.
CTX=n sets the number of cached positions."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["Q3_STAMPS"] = "1"
os.environ.setdefault("Q3_LIB", os.path.join(ROOT, "qwen3.c_amd", "build_astamps", "libq3hip.so"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
path = os.path.join(Q.tmp_dir(), "4B.bin"); Q.synth("4B", path)
CTX = int(os.environ.get("CTX", "0"))
m = hip.q3_model_open(path.encode(), max(1024, CTX + 128), 0)
if CTX: hip.q3_kv_fill_random(m, CTX, 5)
hip.q3_debug_stamps.argtypes = [Q.ModelP, C.POINTER(C.c_uint64), C.c_int]
NWG = 512
tok = 9707
for pos in range(CTX, CTX + 40):
    buf = (C.c_uint64 * (8 * NWG))()
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, 151936)
    if pos - CTX in (5, 20, 39):
        hip.q3_debug_stamps(m, buf, 8 * NWG)
        a = np.array(buf[:], dtype=np.int64).reshape(NWG, 8)
        if pos >= 1024:      # k_merge_wo: 64 one-wave merge workgroups (marks 0 entry, 1 partials in, 2 sums done, 3 stored) + consumers
            mg = np.array([w for w in range(64) if a[w, 3] >= a[w, 0] > 0])
            con = np.array([w for w in range(64, 256) if a[w, 6] >= a[w, 0] > 0])
            t0 = min(a[mg, 0].min(), a[con, 0].min())
            print(f"pos {pos}: {len(mg)} merge + {len(con)} consumer workgroups (k_merge_wo); ns after the first entry")
            rm = (a[mg, :4] - t0) * 10
            for i in range(4):
                print(f"  merge     mark {i}: min {rm[:, i].min():6d}  mean {rm[:, i].mean():8.0f}  max {rm[:, i].max():6d}")
            rc = (a[con, :7] - t0) * 10
            for i in range(7):
                print(f"  consumer  mark {i}: min {rc[:, i].min():6d}  mean {rc[:, i].mean():8.0f}  max {rc[:, i].max():6d}")
            continue
        nslots = 1 if pos < 64 else 16
        n_att = 8 * nslots
        att = np.array([w for w in range(n_att) if (w // 8) <= pos // 64 and a[w, 7] >= a[w, 0] > 0])
        con = np.array([w for w in range(n_att, 256) if a[w, 6] >= a[w, 0] > 0])
        t0 = min(a[att, 0].min(), a[con, 0].min())
        print(f"pos {pos}: {len(att)} attention + {len(con)} consumer workgroups; ns after the first entry")
        ra = (a[att] - t0) * 10
        for i in range(8):
            print(f"  attention mark {i}: min {ra[:, i].min():6d}  mean {ra[:, i].mean():8.0f}  max {ra[:, i].max():6d}")
        rc = (a[con, :7] - t0) * 10
        for i in range(7):
            print(f"  consumer  mark {i}: min {rc[:, i].min():6d}  mean {rc[:, i].mean():8.0f}  max {rc[:, i].max():6d}")
