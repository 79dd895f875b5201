#!/usr/bin/env python3
"""What one RCCL hop costs on this box: the on-device greedy loop of Qwen3-4B shapes (q3_pipeline_run, world 1) with and
without the one-rank self-exchange (Q3_PIPE_SELF=1: a grouped ncclSend + ncclRecv of dim + 1 floats to itself on the launch
stream after every tick).  usage: python tools/bench_rccl_self.py [model] [tokens]"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import q3lib as Q
    hip = Q.hip_lib()
    mdl, n = sys.argv[2], int(sys.argv[3])
    if os.environ.get("Q3_PIPE_SELF") == "1":
        buf = (C.c_char * 128)()
        assert hip.q3_pipeline_unique_id(buf) == 0
        assert hip.q3_pipeline_init(0, 1, bytes(buf)) == 0
    path = os.path.join(Q.tmp_dir(), f"{mdl}.bin"); Q.synth(mdl, path)
    m = hip.q3_model_open(path.encode(), 2048, 0)
    hip.q3_pipeline_run(m, 9707, 0, 16); hip.q3_device_sync(m)
    t0 = time.perf_counter()
    hip.q3_pipeline_run(m, 9707, 16, n); hip.q3_device_sync(m)
    dt = time.perf_counter() - t0
    out = (C.c_int * n)(); hip.q3_pipeline_tokens(m, 0, out, n)
    print(json.dumps({"tok_s": n / dt, "us_per_token": 1e6 * dt / n, "tokens_head": list(out)[:8]}))
    hip.q3_model_close(m)
    sys.exit(0)
mdl = sys.argv[1] if len(sys.argv) > 1 else "4B"
n = sys.argv[2] if len(sys.argv) > 2 else "256"
res = {}
for self_x in ("0", "1"):
    env = dict(os.environ, Q3_PIPE_SELF=self_x, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, __file__, "child", mdl, n], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED", p.stderr[-800:]); sys.exit(1)
    res[self_x] = json.loads(line[-1])
a, b = res["0"], res["1"]
print(f"{mdl}, {n} tokens on the device: plain loop {a['tok_s']:.1f} tok/s ({a['us_per_token']:.1f} us/token); with a one-rank RCCL "
      f"send+recv per tick {b['tok_s']:.1f} tok/s ({b['us_per_token']:.1f} us/token): {b['us_per_token'] - a['us_per_token']:.1f} us per hop; "
      f"same tokens: {a['tokens_head'] == b['tokens_head']}")
