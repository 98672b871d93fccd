set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6a
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_$i.json 2> $O/bench_$i.err; tail -1 $O/bench_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['speedup_vs_vanilla'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline_gemm']['gemm_ms_per_round'])"
done
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $O/bench_cfg1.json 2> $O/bench_cfg1.err; tail -1 $O/bench_cfg1.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['speedup_vs_vanilla'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
