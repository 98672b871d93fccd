# round 6, final pass: smoke, the end-of-milestone profile (bench, rocprofv3 kernel stats, HBM traffic of the two roofline kernels),
# the SQ counter pass of the round's kernels, the other BASELINE configs, three repeats of the headline on this box
set -x
cd $GRAFT_REPO_ROOT
export OUT=gpurun_out/r6final
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
SKIP_TESTS=1 bash tools/round_profile.sh
cd $GRAFT_REPO_ROOT
bash tools/pmc_round_kernels.sh > $OUT/pmc_round_kernels.log 2>&1
cp gpurun_out/pmc_round_kernels.json $OUT/ 2>/dev/null
cd $GRAFT_REPO_ROOT
for c in 0 1 3 4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
timeout 600 python bench.py --config 3 --method seq --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $OUT/bench_cfg3_seq.json 2> $OUT/bench_cfg3_seq.err
timeout 600 python bench.py --shard-path --prefix-per-gpu 16384 --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round > $OUT/bench_shard16k.json 2> $OUT/bench_shard16k.err
for r in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round | tail -1 >> $OUT/bench_repeat.jsonl
done
ls -la $OUT
