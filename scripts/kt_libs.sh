#!/bin/bash
# Per-launch kernel durations of one profiling script under each library given (developer tool):
#   scripts/kt_libs.sh "<script.py> [args...]" lib1.so lib2.so ...
what=$1; shift
for lib in "$@"; do
  echo "== $lib"
  RAYTRACING_AMD_LIB=${GRAFT_REPO_ROOT:-/root/repo}/$lib bash scripts/kt_any.sh $what || exit 1
done
