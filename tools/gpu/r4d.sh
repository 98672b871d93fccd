set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_tail.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4d/pytest_tail.log
timeout 900 python -m pytest tests/test_gpu_generate.py tests/test_harness.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4d/pytest_generate.log
for v in tail notail tail notail; do
  if [ $v = tail ]; then a=""; else a="--no-layer-tail"; fi
  timeout 600 python bench.py --steps 20 --warmup 5 $a 2>gpurun_out/r4d/bench_$v.err | tail -1 >> gpurun_out/r4d/bench_$v.json
done
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 2>gpurun_out/r4d/bench_cfg1.err | tail -1 > gpurun_out/r4d/bench_cfg1.json
