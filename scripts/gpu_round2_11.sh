#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "megakernel comparison" "persistent comparison" "megakernel brute" "persistent brute" "auto comparison"; do
  set -- $cfg
  echo "== $cfg"
  timeout -k 10 300 python bench.py --kernel $1 --mode $2 --spp 64 --cpu-seconds 0 --no-roofline-pass --steps 2 2>/dev/null | cut -c70-200
done
