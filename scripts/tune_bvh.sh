#!/bin/bash
# Walk-slice tuning on the 1.18 M-triangle scene with a -DRT_TUNE build (developer tool)
export RAYTRACING_AMD_LIB=build_ab/_rt_hip_tune.so
for cfg in "8 32 2" "8 16 1" "8 32 1" "8 40 2" "8 32 3" "8 48 3" "12 32 2" "6 32 2" "10 32 2" "8 24 2" "8 0 0"; do
  set -- $cfg
  echo -n "slice $1 more_below $2 times $3: "
  RT_TUNE_SLICE=$1 RT_TUNE_MORE_BELOW=$2 RT_TUNE_MORE_TIMES=$3 timeout -k 5 120 python scripts/prof_bvh.py 6 32 | grep -a "^tris" || exit 1
done
