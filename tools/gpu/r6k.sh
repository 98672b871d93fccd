set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/${OUTNAME:-r6k}
mkdir -p $O
for i in 1 2 3; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round | tail -1 >> $O/bench_repeat.jsonl
done
python - <<PY
import json
for l in open("$O/bench_repeat.jsonl"):
    d=json.loads(l); print(d["value"], d["ms_per_step"], d["speedup_vs_vanilla"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
PY
