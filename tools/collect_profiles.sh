#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics and the two HBM-traffic counter passes of the
# default bench workload (eager launches: rocprofv3 cannot follow this build's graph replays), plus
# the kernel statistics of the same decode step at 4096 cached positions.
# usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
[ -n "$GRAFT_REPO_ROOT" ] || OUT=/root/repo/gpurun_out/prof_$TAG
REPO=$(dirname "$OUT")/..
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp Q3_GRAPH=0
B="--no-cpu-baseline --no-roofline --no-sweep --no-dropin --no-models"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o k --output-format csv -- python3 "$REPO/bench.py" --steps 64 --warmup 8 $B > "$OUT/stats.log" 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o f --output-format csv -- python3 "$REPO/bench.py" --steps 16 --warmup 4 $B > "$OUT/fetch.log" 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o w --output-format csv -- python3 "$REPO/bench.py" --steps 16 --warmup 4 $B > "$OUT/write.log" 2>&1
echo write done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats4096" -o k --output-format csv -- python3 "$REPO/bench.py" --context 4096 --steps 64 --warmup 8 $B > "$OUT/stats4096.log" 2>&1
echo stats4096 done
# keep what profiles/ needs (kernel statistics, counter rows); traces and databases stay on the box (gpurun merges <= 64 MiB)
find "$OUT" -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" -delete
du -sh "$OUT" | tail -1
