set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
L=$PWD/longspec_amd/_lib
timeout 900 python -m pytest tests/test_gpu_tail.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4h/pytest_tail.log
export LONGSPEC_HIP_LIB=$L/liblongspec_hip_hr8.so
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "verify or full_size or prefix or mutant" 2>&1 | tail -12 > gpurun_out/r4h/pytest_ops_hr8.log
for a in "" "--score-scale 4" "--hot-keys 8" "--hot-keys 64" "--sink"; do
  timeout 300 python tools/bench_attn.py --L 16384 131072 --round-like 64 --iters 30 $a >> gpurun_out/r4h/attn_tail_h8.log 2>&1
done
unset LONGSPEC_HIP_LIB
for cfg in 4; do
for ag in 0.01 0.005; do
  timeout 900 python bench.py --config $cfg --agreement $ag --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4h/bench_cfg${cfg}_ag$ag.err | tail -1 >> gpurun_out/r4h/bench_cfg${cfg}_ag$ag.json
done
done
