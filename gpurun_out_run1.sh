set -e
mkdir -p gpurun_out
python __graft_entry__.py smoke
python bench.py --steps 1 --warmup 1 --spp 16 --cpu-seconds 3 | tee gpurun_out/bench_16spp.json
