#!/bin/bash
# On the GPU box: the default bench line, then the rocprofv3 kernel-trace summary and the PMC passes of the same command.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-run}
mkdir -p $R/gpurun_out
cd $R
python bench.py | tee $R/gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_kernel $R/gpurun_out/prof_${TAG}_fetch $R/gpurun_out/prof_${TAG}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kernel -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_kernel.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cat $R/gpurun_out/prof_${TAG}_kernel/*/*kernel_stats.csv | cut -c1-180
