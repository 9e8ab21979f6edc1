# one measurement pass on the GPU box: bench (+e2e), rocprofv3 kernel stats, PMC fetch/write passes, phase stamps
#   gpurun -- 'TAG=r01g bash tools/measure_round.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${TAG:-r01}
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --e2e > gpurun_out/${TAG:-r01}/bench.json 2> gpurun_out/${TAG:-r01}/bench.err
tail -c 600 gpurun_out/${TAG:-r01}/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG:-r01}/prof -o runc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG:-r01}/prof_bench.json 2> gpurun_out/${TAG:-r01}/prof.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG:-r01}/pmc_fetch -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG:-r01}/pmc1.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG:-r01}/pmc_write -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG:-r01}/pmc2.err
FPC_DECODE_STAMPS=1 timeout -k 10 300 python tools/stamp_probe.py 256 2>&1 | grep -v amdgpu.ids | tail -13 > gpurun_out/${TAG:-r01}/stamps.txt
timeout -k 10 300 python tools/voiced_probe.py 256 >> gpurun_out/${TAG:-r01}/stamps.txt 2>&1
ls gpurun_out/${TAG:-r01}/prof
