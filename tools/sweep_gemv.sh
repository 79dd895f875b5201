cd /root/repo
for lib in product pre8 pre16; do for e in 0 1; do
  if [ $lib = product ]; then unset Q3_LIB; else export Q3_LIB=/root/repo/qwen3.c_amd/build_$lib/libq3hip.so; fi
  echo "== lib=$lib early8=$e"; Q3_GEMV_EARLY8=$e timeout -k 10 120 python tools/diag_gemv_loop.py | grep -v "^lib" || exit 1
done; done
