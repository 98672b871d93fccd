set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5f
mkdir -p $O
Lb=$PWD/longspec_amd/_lib
for v in wsprof wsprofpf wsprof wsprofpf; do
  for LL in 131072 16384; do
    echo "== $v L=$LL" >> $O/wsprof.log
    L=$LL LONGSPEC_HIP_LIB=$Lb/liblongspec_hip_$v.so timeout 200 python tools/ws_prof.py >> $O/wsprof.log 2>>$O/wsprof.err
  done
done
cat $O/wsprof.log
