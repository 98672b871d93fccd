set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_topk.py -x -q > gpurun_out/r4w/dist.log 2>&1
tail -3 gpurun_out/r4w/dist.log
timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4w/shard16k.json 2> gpurun_out/r4w/shard16k.err
timeout 600 python bench.py --shard-path --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4w/shard128k.json 2> gpurun_out/r4w/shard128k.err
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --prefix-total 32768 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4w/two_ranks.json 2> gpurun_out/r4w/two_ranks.err
tail -c 400 gpurun_out/r4w/two_ranks.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4w/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('exchange_us_per_call'), d['roofline']['avg_launch_us'], d.get('exchange_timed_out'), d.get('exchange','')[:40])
    except Exception as e: print(f, 'ERR', e)
PY
