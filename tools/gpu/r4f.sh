set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_gpu_tail.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4f/pytest_tail.log
for c in "1 1" "4 2" "1 2" "2 2" "4 4"; do
set -- $c
LS_TAIL_N1_ROWS=$1 LS_TAIL_N2_ROWS=$2 LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_tailprof.so timeout 600 python tools/tail_prof.py > gpurun_out/r4f/tail_prof_n$1_$2.log 2>&1
done
for c in "1 1" "4 2" "1 2"; do
set -- $c
LS_TAIL_N1_ROWS=$1 LS_TAIL_N2_ROWS=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4f/bench_n$1_$2.err | tail -1 >> gpurun_out/r4f/bench_n$1_$2.json
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-layer-tail 2>gpurun_out/r4f/bench_notail.err | tail -1 >> gpurun_out/r4f/bench_notail.json
