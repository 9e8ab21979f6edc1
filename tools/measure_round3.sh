# round-3 measurement pass on the GPU box (bench + e2e, rocprofv3 kernel stats, HBM traffic passes, SQ counters,
# phase stamps, voiced sweep, predictor timing):   gpurun -- 'bash tools/measure_round3.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03m; mkdir -p $o
timeout -k 10 400 python bench.py --steps 20 --warmup 2 > $o/bench.json 2> $o/bench.err
tail -c 400 $o/bench.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err
TAG=r03 bash tools/traffic_round.sh > $o/traffic.log 2>&1
bash tools/pmc_decode.sh r03final > $o/pmc_decode.txt 2>&1
FPC_DECODE_STAMPS=1 timeout -k 10 300 python tools/stamp_probe.py 256 2>&1 | grep -v amdgpu.ids | tail -16 > $o/stamps.txt
timeout -k 10 300 python tools/voiced_probe.py 256 2>&1 | grep -v amdgpu.ids >> $o/stamps.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/enc_prof -o run -- python3 tools/time_encode_split.py > $o/encode_timing.txt 2> $o/enc.err
grep -v amdgpu $o/encode_timing.txt
ls $o/prof $o/enc_prof
