mkdir -p gpurun_out/prof1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r01a -- python bench.py --steps 5 --warmup 1 --cpu-sample-views 0 --stage-iters 5 > gpurun_out/prof1/bench.log 2>&1
ls -R gpurun_out/prof1 | head -30
