set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6d
mkdir -p $O
for i in 1 2 3; do
for v in timed notimed; do
  F=""; [ $v = notimed ] && F="--no-kernel-timing"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla $F > $O/${v}_$i.json 2> $O/${v}_$i.err
  tail -1 $O/${v}_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'])"
done
done
