#!/bin/bash
# A/B timing of libraries on the room and the 1.18 M-triangle scene, best of a few runs each (developer tool):
#   scripts/ab.sh lib1.so lib2.so ...        (ROOM=0 or BVH=0 skips a workload)
for lib in "$@"; do
  [ "${ROOM:-1}" = 1 ] && { RAYTRACING_AMD_LIB=$lib timeout -k 5 150 python scripts/perf.py || exit 1; }
  [ "${BVH:-1}" = 1 ] && { RAYTRACING_AMD_LIB=$lib SPP=${SPP:-32} timeout -k 5 200 python scripts/perf_bvh.py 6 | sed "s|^|$lib |" || exit 1; }
done
exit 0
