set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_ops.py -q -m gpu -k "tail or mutant" 2>&1 | tail -5 > gpurun_out/r4j/pytest_tail_mutant.log
timeout 900 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4j/shard16k.err | tail -1 > gpurun_out/r4j/bench_shard_path_16k.json
timeout 900 python bench.py --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4j/plain16k.err | tail -1 > gpurun_out/r4j/bench_plain_16k.json
timeout 900 python bench.py --shard-path --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4j/shard128k.err | tail -1 > gpurun_out/r4j/bench_shard_path_128k.json
for c in 0 1 3 4; do
timeout 900 python bench.py --config $c --steps 20 --warmup 5 2>gpurun_out/r4j/cfg$c.err | tail -1 > gpurun_out/r4j/bench_cfg$c.json
done
timeout 900 python bench.py --config 3 --method seq --steps 20 --warmup 5 2>gpurun_out/r4j/cfg3seq.err | tail -1 > gpurun_out/r4j/bench_cfg3_seq.json
timeout 900 python bench.py --gpus 2 --backend gloo --share-gpu --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/r4j/gpus2.err | tail -1 > gpurun_out/r4j/bench_gpus2_shared.json
