for a in abl1; do
HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/lib_$a/libhistorian_hip.so timeout -k 10 120 python bench.py --no-cpu-baseline --single-mode --mode fast > gpurun_out/$a.log 2>&1
echo $a $(grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/$a.log)
done
timeout -k 10 120 python bench.py --no-cpu-baseline --single-mode --mode fast > gpurun_out/base.log 2>&1
echo base $(grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/base.log)
