set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train" 2>&1 | tail -2
for v in prev savelate prev savelate prev savelate; do
  export FPC_LIB_PATH=build_variants/lib_$v.so
  echo -n "variant $v: "; timeout -k 10 200 python tools/time_train.py 2>&1 | grep "train step"
done
