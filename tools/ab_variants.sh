for v in "" nostore nolog norot noy noedge nostorelog; do
  for mode in trunc linear; do
    if [ -z "$v" ]; then unset HX_LIB_PATH; else export HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/lib_w$v/libhistorian_hip.so; fi
    echo -n "variant [${v:-product}] "; HX_BENCH_NOCHECK=1 HX_BAND2_NW=4 timeout -k 5 120 python tools/band_quick.py 4096 $mode 20 2000 5 | cut -c1-100
  done
done
