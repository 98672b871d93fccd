set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
L=$PWD/longspec_amd/_lib
export LONGSPEC_HIP_LIB=$L/liblongspec_hip_xcd.so
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4p/pytest_ops_xcd.log
unset LONGSPEC_HIP_LIB
for v in default xcd default xcd; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4p/bench_$v.err | tail -1 >> gpurun_out/r4p/bench_$v.json
  timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4p/bench16_$v.err | tail -1 >> gpurun_out/r4p/bench16_$v.json
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default xcd; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p/rp_$v -- python $R/bench.py --config 1 --steps 12 --warmup 3 --no-vanilla --no-cpu-baseline > $R/gpurun_out/r4p/rp_$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/r4p/rp_$v -name "*.db" | head -1) > $R/gpurun_out/r4p/kernel_stats_16k_$v.csv
  rm -rf $R/gpurun_out/r4p/rp_$v
done
