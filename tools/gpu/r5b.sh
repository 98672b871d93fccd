# round 5, pass b: soak test, ablations of the warp-specialised kernel at the in-round clock (item 5), per-pass launch timeline of
# the round (item 3), QwQ 20k generation (item 4)
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5b
mkdir -p $O
bash tools/gpu/r5b0.sh
timeout 900 python -m pytest tests/test_gpu_generate.py -m gpu -x -q -s -k "soak" > $O/pytest_soak.log 2>&1
tail -6 $O/pytest_soak.log
L=$PWD/longspec_amd/_lib
for rep in 1 2; do
for v in default abl1 abl2 abl4 abl6 abl8 abl16 abl14 abl30 abl25 abl27; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v" >> $O/ablate.log
  timeout 300 python tools/bench_attn.py --L 131072 --round-like 64 --iters 40 >> $O/ablate.log 2>&1
  timeout 300 python tools/bench_attn.py --L 16384 --round-like 64 --iters 60 >> $O/ablate.log 2>&1
done
done
unset LONGSPEC_HIP_LIB
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/rp_trace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-cpu-round --no-vanilla --no-kernel-timing > $O/rp_trace.log 2>&1
python $R/tools/round_timeline.py $(find $O/rp_trace -name "*.db" | head -1) $O/round_timeline_128k.json > $O/round_timeline_128k.txt
cat $O/round_timeline_128k.txt
rm -rf $O/rp_trace
cd $R
timeout 1500 python tools/e2e_longgen.py --model qwq-32b --prompt 32768 --gen 20000 > $O/e2e_qwq_20k.json 2> $O/e2e_qwq_20k.err
tail -c 1500 $O/e2e_qwq_20k.json; tail -5 $O/e2e_qwq_20k.err
