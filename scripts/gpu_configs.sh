#!/bin/bash
# On the GPU box: one bench line per BASELINE.json configuration (C1, C2, C3 whole; C4 and C5 as every rank's share of an 8-GPU job,
# at the stated 1024 / 4096 spp), appended to gpurun_out/configs.jsonl.  usage: gpurun --timeout 1100 -- bash scripts/gpu_configs.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/configs.jsonl
: > $OUT
run() { echo "== $*"; timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>gpurun_out/configs.err | tee -a $OUT | cut -c1-60,230-330 || tail -3 gpurun_out/configs.err; }
timeout -k 10 300 python bench.py --config C1 --steps 20 --warmup 2 --cpu-seconds 5 --no-roofline-pass 2>/dev/null | tee -a $OUT | cut -c1-120
run --config C2 --steps 5
run --config C2 --steps 5 --kernel wavefront
run --config C2 --steps 5 --kernel persistent
run --config C3 --steps 3
for r in 0 1 2 3 4 5 6 7; do run --config C3 --steps 3 --simulate-rank $r --simulate-world 8; done
for r in 0 1 2 3 4 5 6 7; do run --config C4 --steps 2 --simulate-rank $r; done
run --config C4 --steps 1 --gpus 1 --simulate-rank 0 --simulate-world 1
for r in 0 1 2 3 4 5 6 7; do run --config C5 --steps 1 --simulate-rank $r; done
