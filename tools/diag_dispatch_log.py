#!/usr/bin/env python3
"""Prints the AQL dispatch headers (barrier / acquire / release fence scopes) the HIP runtime
uses for the kernels of one decode step, graph replay and eager (AMD_LOG_LEVEL=4)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import q3lib as Q
    hip = Q.hip_lib()
    os.makedirs("/tmp/q3", exist_ok=True)
    path = "/tmp/q3/4Bmini.bin"
    if not os.path.exists(path): Q.synth("4Bmini", path)
    m = hip.q3_model_open(path.encode(), 256, 0)
    tok = 5
    for pos in range(4):
        sys.stderr.write(f"==== step {pos}\n"); sys.stderr.flush()
        lg = hip.forward(m, tok, pos)
    sys.exit(0)
for graph in ("1", "0"):
    env = dict(os.environ, AMD_LOG_LEVEL="4", Q3_GRAPH=graph)
    p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    lines = p.stderr.splitlines()
    idx = [i for i, l in enumerate(lines) if l.startswith("==== step 3")]
    tail = lines[idx[0]:] if idx else lines[-200:]
    print(f"######## Q3_GRAPH={graph}: {len(lines)} log lines, showing dispatch headers of the last step")
    for l in tail:
        if "Header" in l:
            j = l.find("Header")
            print("  ", l[j - 20:j + 70].strip(), "|", (l[l.find("grid="):l.find("grid=") + 60] if "grid=" in l else ""))
