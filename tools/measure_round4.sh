# round-4 measurement pass on the GPU box, every tracked record of the round from ONE tree in ONE call:
#   bash tools/build_variant.sh ws_prof -DFPC_WS_PROF -DFPC_WS_PROF_TAIL      (here, before the call)
#   gpurun --timeout 1200 -- 'bash tools/measure_round4.sh'
# then  python tools/collect_round4.py  copies the summaries into profiles/r04_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04m; rm -rf $o; mkdir -p $o
git rev-parse HEAD > $o/head.txt 2>/dev/null || true
python3 -c "import sys; sys.path.insert(0,'.'); from fpcodec_amd import _lib; print(_lib.lib().fpc_build_info().decode())" > $o/build_info.txt
echo "== bench (20 steps)"; timeout -k 10 500 python bench.py --steps 20 --warmup 2 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.json; echo
echo "== rocprofv3 kernel stats of the bench command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err
echo "== HBM traffic passes (k_decode)"; TAG=r04 bash tools/traffic_round.sh > $o/traffic.log 2>&1; tail -2 $o/traffic.log
echo "== SQ counters (k_decode)"; bash tools/pmc_decode.sh r04final > $o/pmc_decode.txt 2>&1; tail -20 $o/pmc_decode.txt
echo "== phase stamps"; FPC_DECODE_STAMPS=1 timeout -k 10 300 python tools/stamp_probe.py 256 2>&1 | grep -v amdgpu.ids | tail -16 > $o/stamps.txt
timeout -k 10 300 python tools/voiced_probe.py 256 2>&1 | grep -v amdgpu.ids >> $o/stamps.txt
echo "== predictor kernels: rocprofv3 stats + PMC"; TAG=r04 bash tools/profile_encode.sh > $o/profile_encode.log 2>&1; tail -5 $o/profile_encode.log
echo "== predictor kernels: forms, bits and time"; timeout -k 10 400 python tools/ws_check.py > $o/ws_check.txt 2>&1; tail -6 $o/ws_check.txt
echo "== predictor kernels: stage profile (diagnostic build)"
FPC_LIB_PATH=build_variants/lib_ws_prof.so timeout -k 10 300 python tools/ws_prof.py > $o/ws_prof.txt 2>&1 || true
grep -c cycles $o/ws_prof.txt || true
echo "== training step"; timeout -k 10 200 python tools/time_train.py > $o/train.txt 2>&1; tail -2 $o/train.txt
echo "FPC_PRED_WS=0 (the step's forward on the two-role row-split kernel):" >> $o/train.txt
FPC_PRED_WS=0 timeout -k 10 200 python tools/time_train.py 2>&1 | grep "train step" >> $o/train.txt
echo "== chunked decode"; timeout -k 10 300 python tools/chunk_bench.py 0 150 100 50 0 > $o/chunk.txt 2>&1; tail -5 $o/chunk.txt
ls $o
