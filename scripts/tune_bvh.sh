#!/bin/bash
# Walk-slice tuning on the 1.18 M-triangle scene with a -DRT_TUNE build (developer tool):
#   configurations are "slice more_below more_times sub_slice" (wide nodes per trip, lanes-through threshold for another
#   slice, how many more slices, nodes between two rounds of triangle tests)
export RAYTRACING_AMD_LIB=${TUNE_LIB:-variants/_rt_hip_tune.so}
for cfg in "3 32 2 3" "2 32 2 2" "4 32 2 4" "4 32 2 2" "6 32 2 3" "6 32 1 3" "3 32 3 3" "3 48 2 3" "3 16 2 3" "3 0 0 3" "5 32 2 5" "8 32 1 4"; do
  set -- $cfg
  echo -n "slice $1 more_below $2 times $3 sub $4: "
  RT_TUNE_SLICE=$1 RT_TUNE_MORE_BELOW=$2 RT_TUNE_MORE_TIMES=$3 RT_TUNE_SUB_SLICE=$4 timeout -k 5 120 python scripts/prof_bvh.py 6 32 | grep -a "^tris" || exit 1
done
