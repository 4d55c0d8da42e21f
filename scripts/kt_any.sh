#!/bin/bash
# Kernel trace of one profiling script (developer tool): scripts/kt_any.sh <script.py> [args...]  -> per-launch durations of the render kernels
repo=${GRAFT_REPO_ROOT:-/root/repo}
script=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $repo/gpurun_out/kt_any
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/kt_any -o kt --output-format csv -- python3 $repo/scripts/$script "$@" > $repo/gpurun_out/kt_any.log 2>&1
cd $repo
grep -a "set aside\|kernel .* ms rays" gpurun_out/kt_any.log
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/kt_any/kt_kernel_trace.csv")):
    if "k_render" in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:58], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms")
PY
