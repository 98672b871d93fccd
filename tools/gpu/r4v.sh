set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4v
L=$PWD/longspec_amd/_lib
for rep in 1 2; do
for v in default poll_relaxed; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4v/shard16k_${v}_$rep.json 2> gpurun_out/r4v/shard16k_${v}_$rep.err
done
done
export LONGSPEC_HIP_LIB=$L/liblongspec_hip_poll_relaxed.so
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r4v/dist.log 2>&1
tail -3 gpurun_out/r4v/dist.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4v/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('exchange_us_per_call'), d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
