set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6h
mkdir -p $O
timeout 900 python bench.py --gpus 2 --share-gpu --prefix-total 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-cpu-round > $O/bench_2ranks_share.json 2> $O/bench_2ranks_share.err; tail -1 $O/bench_2ranks_share.json | cut -c1-500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --prefix-total 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-cpu-round > $O/bench_2ranks_torchrun.json 2> $O/bench_2ranks_torchrun.err; tail -1 $O/bench_2ranks_torchrun.json | cut -c1-500
