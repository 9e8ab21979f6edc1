# round-6 measurement pass on the GPU box, every tracked record of the round from ONE tree in ONE call:
#   gpurun --timeout 1200 -- 'bash tools/measure_round6.sh'
# then  python tools/collect_round6.py  copies the summaries into profiles/r06_*.
# (the predictor, trainer and k-means kernels are unchanged since round 5: their records stay profiles/r05_*)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06m; rm -rf $o; mkdir -p $o
git rev-parse --short HEAD > $o/head.txt 2>/dev/null || cp build_variants/tree_head.txt $o/head.txt 2>/dev/null || echo "?" > $o/head.txt
python3 -c "import sys; sys.path.insert(0,'.'); from fpcodec_amd import _lib; print(_lib.lib().fpc_build_info().decode())" > $o/build_info.txt
echo "== SQ counters: k_decode, k_decode2, k_encode_wsd (occupancy / issue shares)"
timeout -k 10 900 python3 tools/counters_round.py r06 > $o/counters.log 2>&1 || echo "counters FAILED"; tail -5 $o/counters.log
cp gpurun_out/r06_counters.json profiles/r06_counters.json 2>/dev/null || true   # (bench.py below quotes it: same tree, same hashes)
echo "== HBM traffic passes (k_decode)"; TAG=r06 bash tools/traffic_round.sh > $o/traffic.log 2>&1 || echo "traffic FAILED"; tail -2 $o/traffic.log
cp gpurun_out/r06_traffic.json profiles/r06_traffic.json 2>/dev/null || true
echo "== bench (20 steps)"; timeout -k 10 500 python bench.py --steps 20 --warmup 2 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.json; echo
echo "== rocprofv3 kernel stats of the bench command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err || echo "prof FAILED"
echo "== k_decode2 vs rounds of k_decode: 256 / 384 / 512 / 1024 utterances x 3 s, and half of the frames voiced"
(timeout -k 10 300 python tools/pair_probe.py --T 300 --B 256,384,512,1024; timeout -k 10 200 python tools/pair_probe.py --no-parity --T 300 --B 512 --voiced 0.5) 2>&1 | grep -v amdgpu > $o/pair_probe.txt; cat $o/pair_probe.txt
echo "== phase stamps (diagnostic builds of both kernels, B = 512 x 100 frames)"
for p in -1 1; do echo "-- FPC_LPCNET_PAIRING=$p"; FPC_LPCNET_PAIRING=$p FPC_DECODE_STAMPS=1 timeout -k 10 200 python tools/stamp_probe.py 512 2>&1 | grep -E "phase lengths|wave +[0-9]+ |decode ms" | tail -14; done > $o/stamps.txt; tail -3 $o/stamps.txt
ls $o
