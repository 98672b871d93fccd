set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
L=$PWD/longspec_amd/_lib
timeout 900 python -m pytest tests/test_gpu_linear.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4a/pytest_linear.log
for v in default gemm_nont; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python tools/bench_gemm.py --rows 74 1 > gpurun_out/r4a/bench_gemm_$v.log 2>&1
done
for v in default gemm_nont kvnt default gemm_nont kvnt; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r4a/bench_$v.err | tail -1 >> gpurun_out/r4a/bench_$v.json
done
