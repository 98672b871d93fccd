# round 5, pass j: the long-run GPU tests after the near-tie rule; all bench lines again (the vanilla denominator of pass r5_final
# re-captured its graph every step)
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5j
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_generate.py tests/test_gpu_dist.py -m gpu -q -s > $O/pytest_generate.log 2>&1
grep -a "passed\|failed\|FAILED\|count/num\|rounds reproduce\|soak:" $O/pytest_generate.log | tail -40
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_128k.json 2> $O/bench_128k.err
for c in 0 1 3 4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
done
for r in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round | tail -1 >> $O/bench_repeat.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5j/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['tau'], d.get('vanilla_tokens_per_s'), d.get('speedup_vs_vanilla'), r.get('avg_launch_us'), r.get('frac'))
for l in open('gpurun_out/r5j/bench_repeat.jsonl'):
    d=json.loads(l); print('repeat', d['value'], d['ms_per_step'], d.get('vanilla_tokens_per_s'), d.get('speedup_vs_vanilla'), d['roofline']['avg_launch_us'])
PY
