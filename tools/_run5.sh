cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
for w in 0 1 2 3 4; do
echo "EXP $w"; LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_stamps$w.so timeout 300 python tools/v2_stamps.py run 2>/dev/null | tee gpurun_out/r2e/stamps_exp$w.json | cut -c1-330
done
