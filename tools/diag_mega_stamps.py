import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["Q3_STAMPS"] = "1"
import numpy as np, q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True); path = "/tmp/q3/4B.bin"; Q.synth("4B", path)
m = hip.q3_model_open(path.encode(), 1024, 0)
names = ["x ready","quantised","workers QKV","grid A","attn done","grid B","att gathered","workers Wo","grid C","x gathered","quantised","workers GU","grid D","h gathered","quantised h","workers down","grid E"]
tok = 9707
for pos in range(12):
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, 151936)
    if pos in (3, 11):
        buf = (C.c_uint64 * 64)(); hip.q3_debug_stamps(m, buf, 64)
        t = [buf[i] for i in range(17)]
        print(f"pos {pos}: layer time {(t[16]-t[0])*10/1000:.1f} us")
        for i in range(1, 17):
            print(f"   {names[i]:14s} +{(t[i]-t[i-1])*10/1000:6.2f} us")
