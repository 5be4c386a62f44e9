for n in 0 1 2 3 4 5; do
  if [ $n = 0 ]; then unset HX_LIB_PATH; else export HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/lib_abl$n/libhistorian_hip.so; fi
  echo "== ablate $n"
  timeout -k 10 120 python tools/dag_bench.py 32 fwdonly 2>&1 | grep -E "band|exact |fast " 
done
