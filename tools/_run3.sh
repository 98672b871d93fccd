cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for w in 0 3; do
LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_stamps$w.so timeout 300 python tools/v2_stamps.py run > gpurun_out/r2c/stamps$w.json 2> gpurun_out/r2c/stamps$w.err
cat gpurun_out/r2c/stamps$w.json; tail -3 gpurun_out/r2c/stamps$w.err
done
L=16384 LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_stamps0.so timeout 300 python tools/v2_stamps.py run
