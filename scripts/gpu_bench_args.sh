#!/bin/bash
# On the GPU box: run bench.py once per argument string and print value / ms / roofline for each, e.g.
#   gpurun -- bash scripts/gpu_bench_args.sh "--level 8 --spp 64" "--level 6 --material 1"
# An argument string may start with VAR=value settings (CGPT_WF_* tuning knobs).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "$@"; do
  envs=""; rest=""
  for w in $cfg; do
    if [[ -z "$rest" && "$w" == *=* && "$w" != --* ]]; then envs="$envs $w"; else rest="$rest $w"; fi
  done
  echo "== $cfg"
  env $envs timeout -k 10 500 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 $rest 2>gpurun_out/bench_args.err | tee -a gpurun_out/bench_args.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', 'B/ray', r['algorithmic_bytes_per_ray'], 'frac', r['frac'], 'pipe', r['pipeline_frac'], r['kernel'], 'ms/launch', r['kernel_ms_per_launch'])" || tail -5 gpurun_out/bench_args.err
done
