cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/*
for c in "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc/$tag -o r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc/$tag.err || echo "FAILED $c"
  grep -h "k_decode" gpurun_out/pmc/$tag/r_counter_collection.csv 2>/dev/null | awk -F, '{print $(NF-3), $(NF-2)}' | sort | uniq -c | head -4
done
