"""Turn the passes of scripts/collect_profiles.sh into the counter files bench.py reads (developer tool):
    python scripts/compose_counters.py <tag> [round]   -> profiles/<round>_room_counters.json, _spheres_counters.json,
                                                          _bvh_counters.json, _bench_kernel_stats.csv   (round: r3)
Counters are summed over the chip and over all kernels of the one rt_render call each profiled script makes."""
import csv, glob, json, os, re, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r3"
root = f"gpurun_out/prof_{tag}"


def counters(name):
    """{counter: value} summed over every k_render dispatch of pass `name` (one rt_render call: the lean kernel,
    the full kernel over the set-aside paths and over the given-up jobs) plus the fold kernel."""
    files = glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"] or "k_reduce_" in r["Kernel_Name"]]
    out = {}
    for r in rows:
        out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def log_line(name, pattern):
    m = re.search(pattern, open(f"{root}/{name}.log").read())
    return m


def compose(prefix, workload, kernel, rays_pattern):
    c = {}
    for p in ("fetch", "write", "sq_a", "sq_b", "sq_c", "tc_a"):
        c.update(counters(f"{prefix}_{p}"))
    m = log_line(f"{prefix}_sq_a", rays_pattern)
    kernel_ms, rays = float(m.group(1)), float(m.group(2))
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        d["hbm_bytes (2 * FETCH_SIZE + WRITE_SIZE, KB units; gfx950 tallies 128-byte reads at 64; Infinity-Cache hits included)"] = hbm
        d["hbm_bytes_per_ray"] = hbm / rays
    if "SQ_INSTS_VALU" in c:
        it = rays / 64.0
        d["valu_instructions_per_64_path_segments"] = c["SQ_INSTS_VALU"] / it
        d["salu_instructions_per_64_path_segments"] = c["SQ_INSTS_SALU"] / it
        d["smem_instructions_per_64_path_segments"] = c["SQ_INSTS_SMEM"] / it
        ex = (2.0 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_TRANS_F32"]) * 64.0
        d["executed_fp32_flops ((ADD + MUL + TRANS + 2 FMA) x 64 lanes, masked lanes included)"] = ex
        d["executed_fp32_flops_per_ray"] = ex / rays
        d["instruction_mix"] = {k: c[f"SQ_INSTS_VALU_{k}"] / c["SQ_INSTS_VALU"] for k in ("FMA_F32", "MUL_F32", "ADD_F32", "TRANS_F32", "INT32") }
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_INSTS_VALU" in c:
        d["active_lanes_per_valu_instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64)"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64.0
        d["active_lanes_per_valu_instruction"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64.0
    if "SQ_INSTS_VALU" in c:
        # how hard the vector units are driven: wave-instructions per second over the whole chip, to be read against
        # the rates microbench/valu_rates.hip measures (profiles/r2_valu_rates.txt: 835-965 G/s for full-rate
        # instructions, 515-590 G/s for half-rate ones, ~680 G/s for this kernel's half-and-half mix)
        d["valu_wave_instructions_per_s"] = c["SQ_INSTS_VALU"] / (kernel_ms * 1e-3)
        d["salu_wave_instructions_per_s"] = c["SQ_INSTS_SALU"] / (kernel_ms * 1e-3)
    if "SQ_INSTS_VALU_FMA_F64" in c:
        d["fp64_share_of_valu_instructions"] = (c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]) / c["SQ_INSTS_VALU"]
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        d["l1_miss_rate (TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)"] = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    d["rays"] = rays
    d["kernel_ms (under the counters)"] = kernel_ms
    out = {"workload": workload, "kernel": kernel,
           "collected": "rocprofv3 --pmc, one run per pass (FETCH_SIZE; WRITE_SIZE; two or three passes of SQ counters; cache counters), "
                        "gfx950, scripts/collect_profiles.sh; summed over every kernel of the one rt_render call (lean kernel, full kernel over set-aside paths and given-up jobs, fold)",
           "counters": c, "derived": d}
    out.update({k: d[k] for k in ("hbm_bytes_per_ray", "executed_fp32_flops_per_ray", "active_lanes_per_valu_instruction") if k in d})
    return out


os.makedirs("profiles", exist_ok=True)
jobs = [("room", "scripts/prof_room.py room 256: room 1920x1080, 256 spp, 8 bounces (bench.py's launch), one launch",
         "rt::k_render<1,true,true,true> (+ full kernel passes, fold)"),
        ("spheres", "scripts/prof_room.py random_spheres_seed7 64: random_spheres (seed 7) 1920x1080, 64 spp, 8 bounces (BASELINE configs[1]), one launch",
         "rt::k_render<0,true,true,true> (+ full kernel passes, fold)"),
        ("bvh", "scripts/prof_bvh.py 6 16: room + 1.18 M-triangle mesh (BASELINE config C4's scene), 1920x1080, 16 spp, 8 bounces, one launch",
         "rt::k_render<2,true,true,true> (+ full kernel passes, fold)")]
for name, workload, kernel in jobs:
    if not os.path.exists(f"{root}/{name}_sq_a.log"):
        continue
    out = compose(name, workload, kernel, r"kernel ([0-9.]+) ms rays ([0-9]+)")
    json.dump(out, open(f"profiles/{rnd}_{name}_counters.json", "w"), indent=1)
    print(name, {k.split(" (")[0]: (round(v, 4) if isinstance(v, float) else v) for k, v in out["derived"].items() if not isinstance(v, dict)})
for f in glob.glob(f"{root}/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"profiles/{rnd}_bench_kernel_stats.csv")
