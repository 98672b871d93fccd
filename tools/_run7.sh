cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
for w in 5 6 7; do
echo "EXP $w"; LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_stamps$w.so timeout 300 python tools/v2_stamps.py run 2>/dev/null | tee gpurun_out/r2g/stamps_exp$w.json | cut -c1-290
done
