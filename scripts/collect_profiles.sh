#!/bin/bash
# Every profile the roofline figures of bench.py rest on, in one go (developer tool; run on the GPU box):
#   bash scripts/collect_profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats over bench.py (kernel durations);
# 2. PMC passes, each a run of its own: HBM read bytes, HBM written bytes, SQ instruction mix, SQ activity -
#    over one 256-spp room launch (the bench workload's launch), one 64-spp launch of the spheres scene (BASELINE
#    configs[1]) and one 16-spp launch of the 1.18 M-triangle scene.
# Results under gpurun_out/prof_<tag>/; scripts/compose_counters.py turns them into profiles/<round>_*.json.
set -e
tag=$1
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
SQ_B="SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
SQ_C="SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAIT_ANY SQ_INSTS_VMEM_WR"
TC_A="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- python3 $repo/bench.py --steps 3 --warmup 1 > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "stats done"
pass() {   # name script args... ; counters in $CNT
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc $CNT -d $out/$name -o pmc --output-format csv -- python3 "$@" > $out/$name.log 2>&1
    echo "$name done"
}
CNT="FETCH_SIZE" pass room_fetch $repo/scripts/prof_room.py room 256
CNT="WRITE_SIZE" pass room_write $repo/scripts/prof_room.py room 256
CNT="$SQ_A" pass room_sq_a $repo/scripts/prof_room.py room 256
CNT="$SQ_B" pass room_sq_b $repo/scripts/prof_room.py room 256
CNT="$SQ_C" pass room_sq_c $repo/scripts/prof_room.py room 256
CNT="FETCH_SIZE" pass spheres_fetch $repo/scripts/prof_room.py random_spheres_seed7 64
CNT="WRITE_SIZE" pass spheres_write $repo/scripts/prof_room.py random_spheres_seed7 64
CNT="$SQ_A" pass spheres_sq_a $repo/scripts/prof_room.py random_spheres_seed7 64
CNT="$SQ_B" pass spheres_sq_b $repo/scripts/prof_room.py random_spheres_seed7 64
CNT="$SQ_C" pass spheres_sq_c $repo/scripts/prof_room.py random_spheres_seed7 64
CNT="FETCH_SIZE" pass bvh_fetch $repo/scripts/prof_bvh.py 6 16
CNT="WRITE_SIZE" pass bvh_write $repo/scripts/prof_bvh.py 6 16
CNT="$SQ_A" pass bvh_sq_a $repo/scripts/prof_bvh.py 6 16
CNT="$SQ_B" pass bvh_sq_b $repo/scripts/prof_bvh.py 6 16
CNT="$SQ_C" pass bvh_sq_c $repo/scripts/prof_bvh.py 6 16
CNT="$TC_A" pass bvh_tc_a $repo/scripts/prof_bvh.py 6 16
