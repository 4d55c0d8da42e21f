#!/bin/bash
# Walk-slice tuning on the 1.18 M-triangle scene with a -DRT_TUNE build (developer tool):
#   configurations are "slice more_below more_times sub_slice" (wide nodes per trip, lanes-through threshold for another
#   slice, how many more slices, nodes between two rounds of triangle tests)
export RAYTRACING_AMD_LIB=${TUNE_LIB:-variants/_rt_hip_tune.so}
for cfg in "4 32 2 4" "4 32 2 2" "5 32 2 5" "6 32 2 6" "6 32 2 3" "8 32 2 4" "4 32 3 4" "4 48 2 4" "4 16 2 4" "4 0 0 4" "3 32 2 3" "4 32 1 4"; do
  set -- $cfg
  echo -n "slice $1 more_below $2 times $3 sub $4: "
  RT_TUNE_SLICE=$1 RT_TUNE_MORE_BELOW=$2 RT_TUNE_MORE_TIMES=$3 RT_TUNE_SUB_SLICE=$4 timeout -k 5 120 python scripts/prof_bvh.py 6 32 | grep -a "^tris" || exit 1
done
