#!/bin/bash
# Builds an experiment variant of the library next to the product build (for A/B runs with scripts/gpu_ab_libs.sh or CGPT_LIB_PATH).
# usage: bash scripts/build_variant.sh <tag> [extra hipcc flags...]   ->  cpugpupathtracing_amd/lib/libcpugpupt_<tag>.so
cd "$(dirname "$0")/.."
TAG=$1; shift
CGPT_LIB_PATH=$PWD/cpugpupathtracing_amd/lib/libcpugpupt_$TAG.so CGPT_EXTRA_HIPCC_FLAGS="$*" python -m cpugpupathtracing_amd.build --force
