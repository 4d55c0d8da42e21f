#!/bin/bash
# PMC pass over one room render (developer tool): usage scripts/pmc_room.sh <tag> [spp]
# Two passes of SQ counters; results under gpurun_out/pmc_<tag>_{a,b}/
set -e
tag=$1; spp=${2:-64}
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 -d $repo/gpurun_out/pmc_${tag}_a -o pmc --output-format csv -- python3 $repo/scripts/prof_room.py room $spp
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_CVT SQ_WAVES GRBM_GUI_ACTIVE -d $repo/gpurun_out/pmc_${tag}_b -o pmc --output-format csv -- python3 $repo/scripts/prof_room.py room $spp
