set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
L=$PWD/longspec_amd/_lib
for v in default kstep0; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python tools/bench_gemm.py --rows 74 > gpurun_out/r4b/bench_gemm_$v.log 2>&1
done
for v in default kstep0 default kstep0; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r4b/bench_$v.err | tail -1 >> gpurun_out/r4b/bench_$v.json
done
unset LONGSPEC_HIP_LIB
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4b/pytest_gpu.log
