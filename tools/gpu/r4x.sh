set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4x
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4x/smoke.log 2>&1; tail -1 gpurun_out/r4x/smoke.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r4x/pytest_gpu.log 2>&1; tail -3 gpurun_out/r4x/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4x/bench.json 2> gpurun_out/r4x/bench.err; tail -c 300 gpurun_out/r4x/bench.json
