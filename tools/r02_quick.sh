#!/bin/bash
# quick GPU check of the vocoder after a kernel change: parity tests of the decode path, then timing
#   gpurun -- 'bash tools/r02_quick.sh [variant names for tools/var_bench.py ...]'
mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lpcnet" > gpurun_out/r02/quick_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/r02/quick_tests.txt
if grep -q "GPU core dump" gpurun_out/r02/quick_tests.txt; then exit 1; fi
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/var_bench.py base "$@" > gpurun_out/r02/quick_bench.txt 2>&1
cat gpurun_out/r02/quick_bench.txt
timeout -k 10 300 python tools/var_bench.py --voiced base "$@" >> gpurun_out/r02/quick_bench.txt 2>&1
tail -n +1 gpurun_out/r02/quick_bench.txt | tail -3
if grep -q "GPU core dump" gpurun_out/r02/quick_bench.txt; then exit 1; fi
