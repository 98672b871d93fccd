# round 6, pass i: per-pass launch timeline of the replayed round on the round-6 build (ls_pass_head on)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/rp_trace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-cpu-round --no-vanilla --no-kernel-timing > $O/rp_trace.log 2>&1
python $R/tools/round_timeline.py $(find $O/rp_trace -name "*.db" | head -1) $O/round_timeline_128k.json > $O/round_timeline_128k.txt
head -60 $O/round_timeline_128k.txt
rm -rf $O/rp_trace
