mkdir -p gpurun_out/r03
for v in "$@"; do
echo "== $v"
FPC_LIB_PATH=$PWD/build_variants/$v timeout -k 10 200 python tools/experiments_r03/run_pprof.py 2>&1 | grep k_forward | awk 'NR%2==0'
done > gpurun_out/r03/pprof.log 2>&1
cat gpurun_out/r03/pprof.log
