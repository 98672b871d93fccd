# round 6, pass g: the whole -m gpu suite + smoke + the driver's bench command on the last build, bench --gpus 2 on one GPU
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6g
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; tail -1 $O/bench_driver_flags.json | cut -c1-400
timeout 900 python bench.py --gpus 2 --share-gpu --prefix-total 32768 --steps 10 --warmup 3 --no-cpu-baseline --no-cpu-round > $O/bench_2ranks_share_gpu.json 2> $O/bench_2ranks_share_gpu.err; tail -1 $O/bench_2ranks_share_gpu.json | cut -c1-600; tail -3 $O/bench_2ranks_share_gpu.err
