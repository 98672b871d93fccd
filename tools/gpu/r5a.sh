# round 5, pass a: the new long-run goldens + saturation-bitmap test on the GPU, the whole -m gpu suite, seed sweep of the
# full-size margins, baseline bench lines (default = 128k, cfg1 = 16k, the sharded path at 16k rows per rank)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "long or saturating_key_behind" > $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
timeout 600 python tools/seed_sweep_margins.py --seeds 8 > $O/seed_sweep.json 2> $O/seed_sweep.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_128k.json 2> $O/bench_128k.err
tail -c 600 $O/bench_128k.json
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_shard16k.json 2> $O/bench_shard16k.err
