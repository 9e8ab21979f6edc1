# round-5 measurement pass on the GPU box, every tracked record of the round from ONE tree in ONE call:
#   bash tools/build_variant_pred.sh ws_prof -DFPC_WS_PROF -DFPC_WS_PROF_TAIL ; bash tools/build_variant_pred.sh bwprof0 -DFPC_WS_PROF ;
#   bash tools/build_variant_pred.sh bwprof256 -DFPC_WS_PROF -DFPC_BW_STAMP_TID=256   (here)
#   gpurun --timeout 1200 -- 'bash tools/measure_round5.sh'
# then  python tools/collect_round5.py  copies the summaries into profiles/r05_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05m; rm -rf $o; mkdir -p $o
cp build_variants/tree_head.txt $o/head.txt 2>/dev/null || echo "?" > $o/head.txt
python3 -c "import sys; sys.path.insert(0,'.'); from fpcodec_amd import _lib; print(_lib.lib().fpc_build_info().decode())" > $o/build_info.txt
echo "== bench (20 steps)"; timeout -k 10 500 python bench.py --steps 20 --warmup 2 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.json; echo
echo "== rocprofv3 kernel stats of the bench command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err
echo "== HBM traffic passes (k_decode)"; TAG=r05 bash tools/traffic_round.sh > $o/traffic.log 2>&1; tail -2 $o/traffic.log
echo "== SQ counters (k_decode)"; bash tools/pmc_decode.sh r05final > $o/pmc_decode.txt 2>&1; tail -20 $o/pmc_decode.txt
echo "== predictor kernels: rocprofv3 stats + PMC"; TAG=r05 bash tools/ws_pmc.sh > $o/ws_pmc.log 2>&1; grep "conflict ratio\|k_encode_wsd(\|k_forward_ws<" gpurun_out/r05/ws_pmc/summary.txt | cut -c1-150
echo "== predictor kernels: forms, bits and time"; timeout -k 10 400 python tools/ws_check.py > $o/ws_check.txt 2>&1; tail -6 $o/ws_check.txt
echo "== predictor kernels: stage profile (diagnostic build)"
FPC_LIB_PATH=build_variants/lib_ws_prof.so timeout -k 10 300 python tools/ws_prof.py > $o/ws_prof.txt 2>&1 || true
grep -c cycles $o/ws_prof.txt || true
echo "== training step"; TAG=r05 bash tools/train_prof.sh > $o/train_prof.log 2>&1; head -8 $o/train_prof.log
for i in 1 2; do timeout -k 10 200 python tools/time_train.py 2>&1 | grep "train step\|CPU oracle" >> $o/train.txt; done
echo "FPC_TRAIN_BWD_ROWSPLIT=1 (back-propagation on the row-split kernel, one utterance per workgroups):" >> $o/train.txt
FPC_TRAIN_BWD_ROWSPLIT=1 timeout -k 10 200 python tools/time_train.py 2>&1 | grep "train step" >> $o/train.txt
echo "stage profile of k_train_bwd_ws (-DFPC_WS_PROF build: every launch followed by a host synchronisation):" >> $o/train.txt
FPC_LIB_PATH=build_variants/lib_bwprof0.so timeout -k 10 200 python tools/time_train.py 2>&1 | grep "k_train_bwd_ws" | tail -1 >> $o/train.txt
FPC_LIB_PATH=build_variants/lib_bwprof256.so timeout -k 10 200 python tools/time_train.py 2>&1 | grep "k_train_bwd_ws" | tail -1 >> $o/train.txt
echo "== training step: counters"; TAG=r05 bash tools/train_pmc.sh > $o/train_pmc.log 2>&1; grep "k_train_bwd_ws" gpurun_out/r05/train_pmc/summary.txt | head -20
cat $o/train.txt
echo "== scalar codebooks: k-means"; (timeout -k 10 300 python tools/time_kmeans.py; timeout -k 10 400 python tools/time_kmeans.py 2000000 256) 2>&1 | grep -v amdgpu > $o/kmeans.txt; cat $o/kmeans.txt
ls $o
