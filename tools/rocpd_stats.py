#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of this ROCm's rocprofv3
when no --output-format is given): calls, mean / median / min / max duration in ns.
usage: rocpd_stats.py <results.db> [name substring]"""
import sqlite3, sys, statistics as st
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, end - start from kernels").fetchall()
by = {}
for n, d in rows:
    if sub in n: by.setdefault(n, []).append(d)
tot = sum(sum(v) for v in by.values())
print(f'{"calls":>7} {"total_us":>10} {"mean_ns":>9} {"median_ns":>9} {"min_ns":>8} {"max_ns":>8} {"pct":>6}  name')
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'{len(v):7d} {sum(v)/1e3:10.1f} {sum(v)/len(v):9.0f} {st.median(v):9.0f} {min(v):8d} {max(v):8d} {100*sum(v)/tot:6.2f}  {n[:110]}')
