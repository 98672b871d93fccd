set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_tail.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4e/pytest_tail.log
for u in 0 4 8; do
LS_TAIL_L2_UNITS=$u LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_tailprof.so timeout 600 python tools/tail_prof.py > gpurun_out/r4e/tail_prof_l2_$u.log 2>&1
done
for u in 0 4 8; do
LS_TAIL_L2_UNITS=$u timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4e/bench_l2_$u.err | tail -1 >> gpurun_out/r4e/bench_l2_$u.json
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-layer-tail 2>gpurun_out/r4e/bench_notail.err | tail -1 >> gpurun_out/r4e/bench_notail.json
for a in "" "--score-scale 4" "--hot-keys 8" "--hot-keys 64"; do
  timeout 300 python tools/bench_attn.py --L 16384 131072 --round-like 64 --iters 30 $a >> gpurun_out/r4e/attn_tail_h4.log 2>&1
done
