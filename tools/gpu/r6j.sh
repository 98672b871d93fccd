# round 6, pass j: the per-rank shape of the 128k configuration at N = 2, 4, 8 (64k / 32k / 16k rows of KV per rank) on one GPU:
# bench.py --shard-path runs ONE rank of the sharded path (three-launch attention call, peer-store exchange with itself, vocabulary
# shard off) next to the plain round at the same length
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6j
mkdir -p $O
for L in 65536 32768 16384; do
  timeout 600 python bench.py --shard-path --prefix-per-gpu $L --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla > $O/shard_$L.json 2> $O/shard_$L.err
  timeout 600 python bench.py --prefix-total $L --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla > $O/plain_$L.json 2> $O/plain_$L.err
  for v in shard plain; do tail -1 $O/${v}_$L.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v $L', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('exchange_us_per_call'))"; done
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-round --no-vanilla > $O/plain_131072.json 2> $O/plain_131072.err
tail -1 $O/plain_131072.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('plain 131072', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
