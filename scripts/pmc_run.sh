#!/bin/bash
# Two PMC passes (SQ instruction mix, then SQ activity) over one profiling script (developer tool):
#   scripts/pmc_run.sh <tag> <script.py> [args...]     results under gpurun_out/pmc_<tag>_{a,b}/
set -e
tag=$1; script=$2; shift 2
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 -d $repo/gpurun_out/pmc_${tag}_a -o pmc --output-format csv -- python3 $repo/scripts/$script "$@"
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $repo/gpurun_out/pmc_${tag}_b -o pmc --output-format csv -- python3 $repo/scripts/$script "$@"
