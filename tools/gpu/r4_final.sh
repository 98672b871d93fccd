set -x
cd $GRAFT_REPO_ROOT
export OUT=gpurun_out/r4final
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
SKIP_TESTS=0 bash tools/round_profile.sh
