set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
L=$PWD/longspec_amd/_lib
for rep in 1 2; do
for v in default oprio1 sprio1 valu3 valu2; do
  if [ $v = default ]; then unset LONGSPEC_HIP_LIB; else export LONGSPEC_HIP_LIB=$L/liblongspec_hip_$v.so; fi
  echo "== $v" >> gpurun_out/r4r/attn.log
  timeout 300 python tools/bench_attn.py --L 131072 --round-like 64 --iters 40 >> gpurun_out/r4r/attn.log 2>&1
done
done
