set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4u
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -k "one_launch or peer_store" > gpurun_out/r4u/dist.log 2>&1
tail -5 gpurun_out/r4u/dist.log
for rep in 1 2; do
for one in 1 0; do
  LS_XCHG_ONE_LAUNCH=$one timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4u/shard16k_one${one}_$rep.json 2> gpurun_out/r4u/shard16k_one${one}_$rep.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4u/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('exchange_us_per_call'), d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
