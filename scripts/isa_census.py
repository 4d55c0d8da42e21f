"""Static census of one kernel's ISA by basic block (developer tool).

    python scripts/isa_census.py kernel.s [--blocks]

Classes follow the issue costs measured by microbench/valu_rates.hip on gfx950 (units of one
full-rate vector instruction): full-rate VALU 1, half-rate VALU 1.65, transcendental 3.2, SALU 1.1,
s_waitcnt / s_nop 0.2, memory instructions 1.
"""
import re, sys, collections

FULL = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_xor_b32", "v_and_b32",
        "v_or_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_accvgpr")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if "f64" in op:
        return "f64"
    if op.startswith(FULL):
        return "full"
    if op.startswith("v_"):
        return "half"
    return "other"


COST = dict(full=1.0, half=1.65, cnd=1.65, lane=1.65, f64=1.65, trans=3.2, salu=1.1, wait=0.2, smem=1.0, vmem=1.0, lds=1.0, other=0.0)


def main():
    path = sys.argv[1]
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = line.strip()
        if t.startswith("."):
            continue
        blocks[cur].append(t)
    tot = collections.Counter()
    rows = []
    for name, ins in blocks.items():
        c = collections.Counter(classify(i.split()[0]) for i in ins)
        tot.update(c)
        br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        rows.append((name, len(ins), c, br))
    if "--blocks" in sys.argv:
        for name, n, c, br in rows:
            cost = sum(COST[k] * v for k, v in c.items())
            print(f"{name:12s} n={n:4d} cost={cost:7.1f} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "  -> " + ", ".join(b.split()[-1] for b in br))
    print("TOTAL", dict(tot), "weighted", sum(COST[k] * v for k, v in tot.items()))


if __name__ == "__main__":
    main()
