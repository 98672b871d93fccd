set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
L=$PWD/longspec_amd/_lib
timeout 900 python -m pytest tests/test_gpu_tail.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4g/pytest_tail.log
for v in default odmalate default odmalate; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 300 python tools/bench_attn.py --L 16384 131072 --round-like 64 --iters 40 >> gpurun_out/r4g/attn_$v.log 2>&1
done
for v in default odmalate default odmalate; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4g/bench_$v.err | tail -1 >> gpurun_out/r4g/bench_$v.json
done
export LONGSPEC_HIP_LIB=$L/liblongspec_hip_odmalate.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "verify or full_size or prefix" 2>&1 | tail -5 > gpurun_out/r4g/pytest_ops_odmalate.log
