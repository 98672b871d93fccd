set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r4i/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4i/rp_stats -- python $R/bench.py --steps 20 --warmup 5 --no-vanilla --no-cpu-baseline > $R/gpurun_out/r4i/rp_stats.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/r4i/rp_stats -name "*.db" | head -1) > $R/gpurun_out/r4i/kernel_stats_rounds_only.csv
rm -rf $R/gpurun_out/r4i/rp_stats
