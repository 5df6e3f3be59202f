set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wf2 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --kernel wavefront > $R/gpurun_out/prof_wf2.log 2>&1
cat $R/gpurun_out/prof_wf2/*/*kernel_stats.csv
