"""Summarise scripts/pmc_room.sh output (developer tool): python scripts/pmc_summary.py <tag> <rays>"""
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
# the profiled script renders 4 spp (warm-up) + N spp: counters are summed over both launches
rays = float(sys.argv[2]) if len(sys.argv) > 2 else None
for k in sorted(tot): print(f"{k:28s} {tot[k]:.4e}")
if rays:
    it = rays / 64
    print(f"per wave-iteration: VALU {tot['SQ_INSTS_VALU']/it:.0f}  SALU {tot['SQ_INSTS_SALU']/it:.0f}  SMEM {tot['SQ_INSTS_SMEM']/it:.0f}  "
          f"TRANS {tot['SQ_INSTS_VALU_TRANS_F32']/it:.1f}  INT {tot['SQ_INSTS_VALU_INT32']/it:.0f}  FMA {tot['SQ_INSTS_VALU_FMA_F32']/it:.0f} MUL {tot['SQ_INSTS_VALU_MUL_F32']/it:.0f} ADD {tot['SQ_INSTS_VALU_ADD_F32']/it:.0f}")
    if tot.get('SQ_THREAD_CYCLES_VALU'):
        print(f"active lanes per VALU instruction: {tot['SQ_THREAD_CYCLES_VALU'] / tot['SQ_INSTS_VALU'] / 64:.3f} of 64")
    if tot.get('SQ_INSTS_VMEM_RD'):
        print(f"per wave-iteration: VMEM_RD {tot['SQ_INSTS_VMEM_RD']/it:.1f}  LDS {tot['SQ_INSTS_LDS']/it:.1f}")
