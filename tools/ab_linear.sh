# A/B of library builds on the same box: every historian_amd/lib* directory in turn, twice
for i in 1 2; do
for a in $(cd $GRAFT_REPO_ROOT/historian_amd && ls -d lib*); do
HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/$a/libhistorian_hip.so timeout -k 10 100 python bench.py --no-cpu-baseline --single-mode --steps 5 > gpurun_out/ab_$a.log 2>&1
echo $a $(grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/ab_$a.log)
done; done
