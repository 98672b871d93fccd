set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r4s/dist.log 2>&1
tail -15 gpurun_out/r4s/dist.log
for one in 1 0; do
  LS_XCHG_ONE_LAUNCH=$one timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4s/shard16k_one$one.json 2> gpurun_out/r4s/shard16k_one$one.err
  tail -c 600 gpurun_out/r4s/shard16k_one$one.err
done
LS_XCHG_ONE_LAUNCH=1 timeout 600 python bench.py --shard-path --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4s/shard128k_one1.json 2> gpurun_out/r4s/shard128k_one1.err
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4s/plain16k.json 2> gpurun_out/r4s/plain16k.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('exchange_us_per_call'), d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
