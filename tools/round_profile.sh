# end-of-milestone GPU pass: tests, bench, kernel stats, HBM-traffic counters of the attention and GEMM kernels
#   gpurun --timeout 2400 -- 'OUT=gpurun_out/r2 bash tools/round_profile.sh'   (SKIP_TESTS=1 / SKIP_PMC=1 to shorten)
R=$GRAFT_REPO_ROOT
OUT=$R/${OUT:-gpurun_out/rp}
mkdir -p $OUT
cd $R
if [ "$SKIP_TESTS" != 1 ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log; fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/rp_stats -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-cpu-round > $OUT/rp_stats.log 2>&1
python $R/tools/rocprof_summary.py $(find $OUT/rp_stats -name "*.db" | head -1) > $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-140
if [ "$SKIP_PMC" != 1 ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/rp_$c -- python $R/bench.py --steps 6 --warmup 2 --no-vanilla --no-cpu-baseline --no-cpu-round > $OUT/rp_$c.log 2>&1
done
F=$(find $OUT/rp_FETCH_SIZE -name "*.db" | head -1); W=$(find $OUT/rp_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/pmc_traffic.py $F $W skinny_gemm_kernel $OUT/traffic_gemm.json
python $R/tools/pmc_traffic.py $F $W attn_partial_ws_kernel $OUT/traffic_attn.json
fi
rm -rf $OUT/rp_stats $OUT/rp_FETCH_SIZE $OUT/rp_WRITE_SIZE
