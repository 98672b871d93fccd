set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5s
mkdir -p $O
timeout 900 python bench.py --gpus 2 --backend gloo --share-gpu --prefix-total 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-cpu-round > $O/bench_2ranks_share.json 2> $O/bench_2ranks_share.err
echo rc=$?
tail -c 1500 $O/bench_2ranks_share.json; tail -5 $O/bench_2ranks_share.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --share-gpu --prefix-total 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-cpu-round > $O/bench_2ranks_torchrun.json 2> $O/bench_2ranks_torchrun.err
echo rc=$?
tail -c 600 $O/bench_2ranks_torchrun.json
