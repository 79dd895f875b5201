#!/usr/bin/env python3
"""Runs bench.py over the BASELINE.json configurations that fit one GPU and prints one
line per run (context sweep on 4B, other model sizes).  Usage: python tools/sweep.py [ctx|models]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print("FAILED", args, p.stderr[-400:])
        return None
    return json.loads(line[-1])


what = sys.argv[1] if len(sys.argv) > 1 else "ctx"
if what == "ctx":
    for ctx in (0, 512, 4096):
        d = run(["--steps", "64", "--warmup", "8", "--context", str(ctx), "--seq-len", "8192", "--no-cpu-baseline"])
        if d:
            k = d.get("kernels", {})
            print(f"4B context {ctx:5d}: {d['value']:8.1f} tok/s  {d['ms_per_step']:.3f} ms  frac {d['hbm_roofline_frac_step']:.3f}  "
                  f"attn {k.get('attn', {}).get('us')} us  combine {k.get('attn_combine', {}).get('us')} us", flush=True)
else:
    for mdl in ("0.6B", "1.7B", "4B", "8B"):
        d = run(["--steps", "128", "--warmup", "8", "--model", mdl, "--no-cpu-baseline", "--no-roofline", "--no-dropin"])
        if d:
            c = d.get("contexts", {})
            ctx = "  ".join(f"ctx {k}: {v['tokens_per_s']:.0f} ({v['frac_of_hbm_roofline']:.3f})" for k, v in c.items() if k != "0")
            print(f"{mdl:5s}: {d['value']:8.1f} tok/s  {d['ms_per_step']:.3f} ms  frac {d['hbm_roofline_frac_step']:.3f}  "
                  f"prefill {d.get('prefill_tokens_per_s')} tok/s  {ctx}", flush=True)
