set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5m
mkdir -p $O
for s in 0 1 2 3; do
  timeout 900 python tools/fuzz_attn.py --cases 250 --seed $s > $O/fuzz_seed$s.log 2>&1
  tail -2 $O/fuzz_seed$s.log
done
