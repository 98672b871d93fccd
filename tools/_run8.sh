cd $GRAFT_REPO_ROOT
for m in prefix verify; do
echo main $m; LS_ATTN_KERNEL=v2 timeout 300 python tools/bench_attn.py --L 131072 --iters 30 --mode $m 2>/dev/null
echo stamps $m; LONGSPEC_HIP_LIB=$PWD/longspec_amd/_lib/liblongspec_hip_stamps0.so LS_ATTN_KERNEL=v2 timeout 300 python tools/bench_attn.py --L 131072 --iters 30 --mode $m 2>/dev/null
done
