#!/bin/bash
# On the GPU box: the round's roofline inputs (single-pool kernel-trace + PMC passes, scripts/gpu_roofline_pmc.sh) for every workload
# bench.py prints a roofline for, installed into profiles/pmc_counts.json with the hash of the device sources they were counted on.
# usage: gpurun --timeout 1100 -- bash scripts/gpu_final_profiles.sh <round> [tags...]      tags: c3 c3pt c2 c4 c5 (default: all)
cd $GRAFT_REPO_ROOT
export INSTALL=${1:-r03}; shift
TAGS=${*:-c3 c3pt c2 c4 c5}
for T in $TAGS; do
  case $T in
    c3)   A="--config C3" ;;
    c3pt) A="--config C3 --kernel persistent" ;;
    c2)   A="--config C2" ;;
    c4)   A="--config C4" ;;
    c5)   A="--config C5" ;;
  esac
  echo "==== $T: $A"
  PASS_TIMEOUT=400 bash scripts/gpu_roofline_pmc.sh $T $A | tail -14
done
mkdir -p gpurun_out/profiles_$INSTALL && cp -r profiles/$INSTALL/roofline_* profiles/pmc_counts.json gpurun_out/profiles_$INSTALL/
