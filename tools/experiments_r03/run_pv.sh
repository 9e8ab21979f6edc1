mkdir -p gpurun_out/r03
{
echo "== shipped, auto (n=2)"; timeout -k 10 120 python tools/time_encode_split.py 2>&1 | head -6
for v in A B C; do for n in 2 4; do
echo "== variant $v FPC_PRED_SPLIT=$n"; FPC_LIB_PATH=$PWD/build_variants/lib_p$v.so FPC_PRED_SPLIT=$n FPC_SPIN_LIMIT_US=200000 timeout -k 10 120 python tools/time_encode_split.py 2>&1 | head -6
done; done
} > gpurun_out/r03/pv.log 2>&1
cat gpurun_out/r03/pv.log
