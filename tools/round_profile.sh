# end-of-milestone GPU pass: tests, bench, kernel stats, HBM-traffic counters of the GEMM kernel
R=$GRAFT_REPO_ROOT
cd $R
if [ "$SKIP_TESTS" != 1 ]; then timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log; fi
timeout 500 python bench.py > gpurun_out/bench_rp.json 2> gpurun_out/bench_rp.err; tail -1 gpurun_out/bench_rp.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rp_stats -- python $R/bench.py --steps 12 --warmup 3 > $R/gpurun_out/rp_stats.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/rp_stats -name "*.db" | head -1) > $R/gpurun_out/rp_kernel_stats.csv
head -12 $R/gpurun_out/rp_kernel_stats.csv | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/rp_$c -- python $R/bench.py --steps 6 --warmup 2 --no-vanilla > $R/gpurun_out/rp_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $(find $R/gpurun_out/rp_FETCH_SIZE -name "*.db" | head -1) $(find $R/gpurun_out/rp_WRITE_SIZE -name "*.db" | head -1) skinny_gemm_kernel $R/gpurun_out/rp_traffic_gemm.json
python $R/tools/pmc_traffic.py $(find $R/gpurun_out/rp_FETCH_SIZE -name "*.db" | head -1) $(find $R/gpurun_out/rp_WRITE_SIZE -name "*.db" | head -1) attn_partial_ws_kernel $R/gpurun_out/rp_traffic_attn.json
rm -rf $R/gpurun_out/rp_stats $R/gpurun_out/rp_FETCH_SIZE $R/gpurun_out/rp_WRITE_SIZE
