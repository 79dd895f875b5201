import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["Q3_STAMPS"] = "1"
if len(sys.argv) > 2: os.environ["HIP_FORCE_DEV_KERNARG"] = sys.argv[2]
os.environ["Q3_GRAPH"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ.setdefault("Q3_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qwen3.c_amd", "build_astamps", "libq3hip.so"))
import numpy as np, q3lib as Q
hip = Q.hip_lib()
os.makedirs("/tmp/q3", exist_ok=True); path = "/tmp/q3/4B.bin"; Q.synth("4B", path)
CTX = int(os.environ.get('CTX', '0'))
m = hip.q3_model_open(path.encode(), max(1024, CTX + 128), 0)
if CTX: hip.q3_kv_fill_random(m, CTX, 5)
hip.q3_debug_stamps.argtypes = [Q.ModelP, C.POINTER(C.c_uint64), C.c_int]
tok = 9707
for pos in range(CTX, CTX + 40):
    lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, 151936)
    if pos - CTX in (0, 1, 5, 20, 39):
        buf = (C.c_uint64 * 16)(); hip.q3_debug_stamps(m, buf, 16)
        rt = [buf[2*i] for i in range(8)]; cy = [buf[2*i+1] for i in range(8)]
        d_rt = [(rt[i]-rt[0])*10 for i in range(8)]   # ns
        d_cy = [cy[i]-cy[0] for i in range(8)]
        clk = (cy[7]-cy[0]) / max(1, (rt[7]-rt[0])*10) 
        print(f"pos {pos}: ns from start {d_rt}  cycles {d_cy}  clock ~{clk:.2f} GHz")
