cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
for v in shipped lib_pold.so; do
  if [ "$v" != shipped ]; then export FPC_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$v; fi
  rm -rf gpurun_out/r03b/tr_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03b/tr_$v -o run -- python3 tools/time_train.py > /dev/null 2>&1
  echo "== $v"; f=$(find gpurun_out/r03b/tr_$v -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | sed 's/(anonymous namespace):://g' | cut -c1-150 | head -9
done
