set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
L=$PWD/longspec_amd/_lib
export LONGSPEC_HIP_LIB=$L/liblongspec_hip_pipe2.so
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "verify or full_size or prefix or saturating or prefill or append" 2>&1 | tail -8 > gpurun_out/r4o/pytest_ops_pipe2.log
unset LONGSPEC_HIP_LIB
for v in default pipe2 default pipe2; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 300 python tools/bench_attn.py --L 16384 131072 --round-like 64 --iters 40 >> gpurun_out/r4o/attn_$v.log 2>&1
done
for v in default pipe2 default pipe2; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4o/bench_$v.err | tail -1 >> gpurun_out/r4o/bench_$v.json
done
