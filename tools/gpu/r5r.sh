set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5r
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dist.py -m gpu -q > $O/pytest_dist.log 2>&1; tail -6 $O/pytest_dist.log
