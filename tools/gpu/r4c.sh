set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
L=$PWD/longspec_amd/_lib
timeout 900 python tools/bench_l2_prefetch.py --rows 74 1 > gpurun_out/r4c/l2_prefetch.log 2>&1
for a in "" "--score-scale 4" "--score-scale 8" "--sink" "--hot-keys 8" "--hot-keys 64"; do
  timeout 300 python tools/bench_attn.py --L 16384 131072 --round-like 64 --iters 30 $a >> gpurun_out/r4c/attn_tail.log 2>&1
done
for v in default kvnt0; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  timeout 900 python bench.py --config 4 --steps 20 --warmup 5 2>gpurun_out/r4c/bench_cfg4_$v.err | tail -1 >> gpurun_out/r4c/bench_cfg4_$v.json
done
unset LONGSPEC_HIP_LIB
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 2>gpurun_out/r4c/bench_cfg1.err | tail -1 > gpurun_out/r4c/bench_cfg1.json
