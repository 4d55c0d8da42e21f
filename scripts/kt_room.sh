#!/bin/bash
# Kernel trace of one 256-spp room launch (developer tool): per-kernel durations
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/kt_room -o kt --output-format csv -- python3 $repo/scripts/prof_room.py ${1:-room} ${2:-256} > $repo/gpurun_out/kt_room.log 2>&1
cd $repo
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/kt_room/kt_kernel_stats.csv")):
    print(r["Name"][:72], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), "ms")
PY
