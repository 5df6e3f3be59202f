set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --kernel wavefront --cpu-seconds 0 | tee gpurun_out/bench_wf_first.json
