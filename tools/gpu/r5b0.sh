set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5b
mkdir -p $O
for r in long_qwen_g5_s0 long_qwen_bf16_g5_s3 long_qwen_g7_s2 long_gqa_s1; do
  timeout 300 python tools/dbg_long_run.py $r >> $O/dbg_long.jsonl 2>> $O/dbg_long.err
done
cat $O/dbg_long.jsonl
