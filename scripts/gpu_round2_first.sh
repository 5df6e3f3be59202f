#!/bin/bash
# first GPU call of round 2: the whole GPU suite, the bench line with the measured issue table, frame times at 1 spp per call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_pytest.log
timeout -k 10 400 python bench.py --issue-table > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; echo "bench rc=$?"; cat gpurun_out/r02_bench_a.json; grep "issue rate" gpurun_out/r02_bench_a.err
timeout -k 10 200 python scripts/gpu_frame_time.py > gpurun_out/r02_frame_time.txt 2>&1; cat gpurun_out/r02_frame_time.txt
timeout -k 10 200 python scripts/gpu_frame_time.py 1920 1080 trace_events=0 > gpurun_out/r02_frame_time_noev.txt 2>&1; grep wavefront gpurun_out/r02_frame_time_noev.txt
