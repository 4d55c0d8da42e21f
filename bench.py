#!/usr/bin/env python3
"""Headline benchmark: rays/s of the per-pixel render path on room.py at 1920x1080, 256 spp,
8 bounces (BASELINE.json configs[2], the configuration its metric is quoted on; it fits one GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one full render of the workload.  With N > 1 the 256 samples of every pixel are split
across the ranks (weak scaling is NOT what this does: total work is fixed, so "scaling": "strong"),
each rank renders its share into its own device buffer, and one RCCL all-reduce merges the
buffers; the timed region contains the render kernels and the all-reduce.  Scene tables and the
sky image are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0, with
  roofline     - float32 VALU roofline of the render kernel: algorithmic FLOPs (tests executed x
                 fixed per-test constants, listed below) / kernel time from HIP events recorded
                 on the launch stream, against the 157.3 TFLOP/s vector peak
  cpu_baseline - the double-precision CPU restatement (oracle/rt_oracle.c) of the same path,
                 timed on this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WIDTH, HEIGHT, SPP, BOUNCES, SEED = 1920, 1080, 256, 8, 2025
PEAK_FP32_VALU_TFLOPS = 157.3
# Algorithmic FLOPs per unit of work (FMA = 2, every other arithmetic op or compare = 1, loop
# invariants hoisted): ray-sphere test 20 (sphere.py:15-20), + 12 when the discriminant is not
# negative, ray-triangle test 46 (mesh.py:122-136 with ab, ac, N precomputed), one bounce 120
# (renderer.py:39-79 + hit point / normal), one camera ray 40 (camera.py:40-49), one sky lookup 20.
F_SPHERE, F_SPHERE_HIT, F_TRI, F_SHADE, F_RAYGEN, F_SKY = 20, 12, 46, 120, 40, 20


def load_room():
    from raytracing_amd.backend.packer import PackedScene

    gold = os.path.join(REPO, "tests", "golden")
    sky = np.ascontiguousarray(np.load(os.path.join(gold, "skybox1_rgb8.npz"))["rgb"])
    return PackedScene.load(os.path.join(gold, "scene_room.npz"), sky_rgb=sky)


def measured_traffic():
    """HBM bytes per launch of the render kernel on THIS workload, from the committed PMC passes
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside a timed benchmark); None if absent."""
    try:
        with open(os.path.join(REPO, "profiles", "r1_hbm_traffic.json")) as fh:
            return float(json.load(fh)["bytes_per_launch"])
    except Exception:
        return None


def algorithmic_flops(st: dict) -> float:
    return (st["sphere_tests"] * F_SPHERE + st["tri_tests"] * F_TRI + st["shaded_hits"] * F_SHADE
            + st["paths"] * F_RAYGEN + st["sky_lookups"] * F_SKY)


def cpu_baseline(scene) -> dict:
    """Time the CPU restatement on a bounded sample of the same workload: the same scene, camera and
    bounce limit at 960x540 (a quarter of the pixels), with as many samples per pixel as make
    roughly 15 s of work on the threads available (at most 256, the workload's own count)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers as h  # the oracle loader lives with the tests: it is test infrastructure

    lib = h.oracle()
    w, hgt = 960, 540
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or lib.orc_max_threads()
    threads = max(1, min(threads, 64))
    # single thread: the faithful analogue of upstream's GIL-bound thread pool
    t0 = time.perf_counter()
    _, rays1 = h.oracle_render(scene, w, hgt, 2, BOUNCES, seed=SEED, nthreads=1)
    dt1 = time.perf_counter() - t0
    rate1 = rays1 / dt1
    spp = int(max(1, min(SPP, (15.0 * rate1 * threads * 0.7) / max(rays1 / 2, 1))))
    t0 = time.perf_counter()
    _, rays = h.oracle_render(scene, w, hgt, spp, BOUNCES, seed=SEED, nthreads=threads)
    dt = time.perf_counter() - t0
    return {"value": rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"room.py {w}x{hgt}, {spp} spp, {BOUNCES} bounces, double precision, {threads} OpenMP threads "
                      f"({rays} rays in {dt:.2f} s)",
            "single_thread_value": rate1, "single_thread_sample": f"{w}x{hgt}, 2 spp ({rays1} rays in {dt1:.2f} s)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # One rank per GPU.  RT_BENCH_BACKEND=gloo (a rehearsal switch) lets several ranks share the
    # GPUs that exist, e.g. two ranks on a one-GPU box, to exercise the sharded path without RCCL.
    backend_name = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend_name != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend_name)

    from raytracing_amd.backend.lib import HipBackend
    from raytracing_amd.distributed import DistributedRenderer

    scene = load_room()
    be = HipBackend(local_rank)
    be.set_scene(scene)
    accum = torch.zeros((WIDTH, HEIGHT, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    dr = DistributedRenderer(be, accum)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        accum.zero_()
        dr.render_local(WIDTH, HEIGHT, SPP, BOUNCES, SEED, stream=stream)
        dr.merge()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("RT_BENCH_TRACE")   # progress lines on stderr, for locating a stall

    def say(msg):
        if trace:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    say("scene set, warming up")
    for _ in range(a.warmup):
        step()
    fence()
    say("warm-up done")
    be.reset_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    say("timed steps done")
    st = be.stats()   # this rank's counters over the timed steps

    # whole-job numbers: sum of rays over ranks, max of time over ranks
    tot = torch.tensor([float(st["rays"]), float(st["paths"]), algorithmic_flops(st), st["kernel_ms"]],
                       dtype=torch.float64, device=f"cuda:{local_rank}")
    tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays_all, paths_all, flops_all, kms_all = [float(x) for x in tot.tolist()]
    dt_all = float(tmax.item())
    say("reduced")

    if rank == 0:
        launches = max(int(st["launches"]), 1)
        kernel_s = st["kernel_ms"] * 1e-3 / launches            # average launch duration, this rank
        flops_launch = algorithmic_flops(st) / launches
        achieved = flops_launch / kernel_s / 1e12
        out = {
            "metric": "rays/sec at 1/2/4/8 MI355X, room.py 1080p; % of FP32 VALU peak",
            "value": rays_all / dt_all,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt_all / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: upstream room scene tables + decoded sky image (tests/golden), Philox sample stream",
            "config": {"workload": "scenes/room.py 1920x1080, 256 spp, 8 bounces, skybox (BASELINE configs[2])",
                       "width": WIDTH, "height": HEIGHT, "spp": SPP, "max_bounces": BOUNCES,
                       "parallelism": f"spp-shard x{world} + all-reduce" if world > 1 else "single GPU"},
            "samples_per_s": paths_all / dt_all,
            "rays_per_path": rays_all / max(paths_all, 1.0),
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": measured_traffic() if world == 1 else None,
                         "traffic_note": "HBM bytes per launch at N=1, PMC passes in profiles/r1_hbm_traffic.json (2*FETCH_SIZE + WRITE_SIZE)",
                         "kernel": "rt::k_render", "kernel_ms_per_launch": kernel_s * 1e3,
                         "flops_per_launch": flops_launch,
                         "exact_ray_fraction": st["exact_rays"] / max(st["rays"], 1),
                         "refined_hit_fraction": st["refined_hits"] / max(st["rays"], 1)},
        }
        if not a.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
