set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4rep
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4rep/b$i.json 2> gpurun_out/r4rep/b$i.err
done
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4rep/cfg1.json 2> gpurun_out/r4rep/cfg1.err
timeout 900 python bench.py --config 4 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r4rep/cfg4.json 2> gpurun_out/r4rep/cfg4.err
timeout 900 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4rep/cfg3.json 2> gpurun_out/r4rep/cfg3.err
timeout 900 python bench.py --config 0 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4rep/cfg0.json 2> gpurun_out/r4rep/cfg0.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4rep/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['tau'], d['roofline']['frac'], d['roofline_gemm']['gemm_ms_per_round'], d.get('speedup_vs_vanilla'))
    except Exception as e: print(f, 'ERR', e)
PY
