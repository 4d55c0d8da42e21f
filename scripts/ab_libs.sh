#!/bin/bash
# Time the room and the 1.18 M-triangle scene with each library given (developer tool): scripts/ab_libs.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  RAYTRACING_AMD_LIB=$lib timeout -k 5 120 python scripts/prof_room.py room 64 | tail -1
  RAYTRACING_AMD_LIB=$lib timeout -k 5 120 python scripts/prof_bvh.py 6 32 | tail -1
done
