set -x
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -3 gpurun_out/bench_r1.err; cat gpurun_out/bench_r1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head -20
