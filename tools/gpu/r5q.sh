set -x
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5q
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/rp -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-cpu-round --no-vanilla --no-kernel-timing > $O/rp.log 2>&1
python $R/tools/round_timeline.py $(find $O/rp -name "*.db" | head -1) $O/round_timeline_128k.json > $O/round_timeline_128k.txt
cat $O/round_timeline_128k.txt
rm -rf $O/rp
