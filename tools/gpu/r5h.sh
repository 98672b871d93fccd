# round 5, pass h: the whole -m gpu suite on the current tree + bench lines
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5h
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -a "passed\|failed\|FAILED\|count/num\|rounds reproduce\|soak:\|one split" $O/pytest_gpu.log | tail -50
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_128k.json 2> $O/bench_128k.err
tail -c 400 $O/bench_128k.json
