# round 5, pass o: per-kernel timeline of the sharded path at 16k rows per rank (stage 1 / reduce+push / wait+merge) next to the plain one
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5o
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in shard plain; do
  if [ $v = shard ]; then A="--shard-path --prefix-per-gpu 16384"; else A="--config 1"; fi
  timeout 600 rocprofv3 --kernel-trace -d $O/rp_$v -- python $R/bench.py $A --steps 12 --warmup 3 --no-cpu-baseline --no-cpu-round --no-vanilla --no-kernel-timing > $O/rp_$v.log 2>&1
  python $R/tools/round_timeline.py $(find $O/rp_$v -name "*.db" | head -1) $O/round_timeline_16k_$v.json > $O/round_timeline_16k_$v.txt
  cat $O/round_timeline_16k_$v.txt
  rm -rf $O/rp_$v
done
