set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "verify_attention_golden or draft_attention_chain_golden" > $O/pytest_attn_golden.log 2>&1; tail -12 $O/pytest_attn_golden.log
