set -e
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r2l
cd /tmp && export TMPDIR=/tmp Q3_GRAPH=0 Q3_LIB=$GRAFT_REPO_ROOT/qwen3.c_amd/build_gdbg/libq3hip.so
for v in 0 1 2 4 3 7; do
  Q3_GEMM_DBG=$v timeout -k 10 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2l/gd$v -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_prefill.py 128 > $GRAFT_REPO_ROOT/gpurun_out/r2l/gd$v.log 2>&1
  echo "dbg $v" >> $GRAFT_REPO_ROOT/gpurun_out/r2l/gd.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/gpurun_out/r2l/gd$v/k_results.db gemm_q8 >> $GRAFT_REPO_ROOT/gpurun_out/r2l/gd.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2l/gd$v
done
