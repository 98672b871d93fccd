cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2i/gpu_tests.log 2>&1; tail -5 gpurun_out/r2i/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r2i/bench_128k.json 2> gpurun_out/r2i/bench_128k.err; tail -c 3000 gpurun_out/r2i/bench_128k.json; tail -5 gpurun_out/r2i/bench_128k.err
