set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 | tee $R/gpurun_out/bench_r01_first.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_kernel -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_r01_kernel.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_r01_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_r01_write.log 2>&1
ls -R $R/gpurun_out | head -50
