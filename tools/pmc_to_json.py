#!/usr/bin/env python3
"""Turns the counter CSVs of tools/collect_profiles.sh into profiles/<tag>_pmc_traffic.json and copies
the kernel statistics next to it.  usage: pmc_to_json.py gpurun_out/prof_<tag> <tag>"""
import csv, glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS = [("k_gemv3<1, 2,", "gateup"), ("k_gemv3<2, 1,", "down"), ("k_gemv3<1, 0,", "qkv"), ("k_gemv3<0, 1,", "wo"),
         ("k_gemv2<", "cls"), ("k_attn_wo<", "attn_wo"), ("k_merge_wo<", "merge_wo"), ("k_attn_long<", "attn_long"),
         ("k_attn<", "attn"), ("k_attn_merge", "attn_merge")]
ALG = {"gateup": 52920320, "down": 26460160, "qkv": 16711680, "wo": 11141120, "cls": 413265920}
def per_kernel(pattern, counter):
    acc = {}
    for f in glob.glob(os.path.join(src, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter: continue
            for key, name in CLASS:
                if key in r["Kernel_Name"]:
                    a = acc.setdefault(name, [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
                    break
    return acc
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) -- python3 bench.py "
                 "--steps 16 --warmup 4 (eager launches), Qwen3-4B shapes, MI355X; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 counts the 128-B requests of wide streaming reads as 64 B)", "kernels": {}}
for name in fetch:
    f_kib = fetch[name][0] / fetch[name][1]
    w_kib = write.get(name, [0.0, 1])[0] / max(1, write.get(name, [0.0, 1])[1])
    k = {"FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB_raw": round(w_kib, 1),
         "hbm_read_bytes_corrected": int(f_kib * 1024 * 2), "hbm_write_bytes": int(w_kib * 1024), "launches": fetch[name][1]}
    if name in ALG:
        k["algorithmic_bytes"] = ALG[name]
        k["traffic_over_algorithmic"] = round((k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"]) / ALG[name], 3)
    out["kernels"][name] = k
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats4096", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_ctx4096_kernel_stats.csv"))
print(json.dumps(out["kernels"], indent=1))
