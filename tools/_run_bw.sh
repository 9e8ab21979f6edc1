set -e
for v in prev xcd xcdnp prev xcd xcdnp; do
  export FPC_LIB_PATH=build_variants/lib_$v.so
  echo -n "variant $v: "; timeout -k 10 200 python tools/time_train.py 2>&1 | grep "train step"
done
unset FPC_LIB_PATH
TAG=r05g bash tools/train_pmc.sh 2>&1 | grep -E "k_grad_tn +(TCC|SQ_WAVE_CYCLES|SQ_WAIT_INST)"
