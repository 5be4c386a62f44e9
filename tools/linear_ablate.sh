# Same-box comparison of an HX_ABLATE build of the library (historian_amd/lib_abl1, see hx_linear.hip) with the product build
for i in 1 2; do
for a in lib_abl1 lib; do
HX_BENCH_NOCHECK=1 HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/$a/libhistorian_hip.so timeout -k 10 100 python bench.py --no-cpu-baseline --single-mode --steps 5 > gpurun_out/ab_$a.log 2>&1
echo $a $(grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/ab_$a.log)
done; done
