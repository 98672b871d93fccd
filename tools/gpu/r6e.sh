# round 6, pass e: end-to-end runs on the round-6 build -- a real 131072-token prompt (prefill + tree / vanilla decode), configs[4]
# as written (QwQ dims, bf16, real 32k prompt, 20 000 generated tokens), 1000 random attention cases
set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6e
mkdir -p $O
timeout 1200 python tools/e2e_fullsize.py --prompt 131072 --gen 256 > $O/e2e_fullsize_128k.json 2> $O/e2e_fullsize_128k.err; tail -c 1200 $O/e2e_fullsize_128k.json
timeout 1500 python tools/e2e_longgen.py --model qwq-32b --prompt 32768 --gen 20000 > $O/e2e_qwq_20k.json 2> $O/e2e_qwq_20k.err; tail -c 1500 $O/e2e_qwq_20k.json; tail -3 $O/e2e_qwq_20k.err
for s in 0 1 2 3; do
  timeout 900 python tools/fuzz_attn.py --cases 250 --seed $s > $O/fuzz_seed$s.log 2>&1
  tail -2 $O/fuzz_seed$s.log
done
