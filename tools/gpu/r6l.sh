set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_generate.py -m gpu -x -q -k "vanilla_step_loses or per_round_bound" > $O/new_tests.log 2>&1; tail -15 $O/new_tests.log
