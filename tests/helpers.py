"""Shared test plumbing: golden fixtures, the CPU oracle (ctypes) and the packed scenes.

The oracle under oracle/ is test infrastructure: it is loaded HERE (and by bench.py's cpu_baseline
leg and __graft_entry__.smoke()), never by the product package.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from functools import lru_cache

import numpy as np

from raytracing_amd.backend.packer import PackedScene, RtSceneDesc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "_build", "librt_oracle.so")

SEED_RENDER = 0x5EED1234ABCD

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


@lru_cache(maxsize=None)
def oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    lib = C.CDLL(ORACLE_SO)
    lib.orc_uniform.restype = C.c_double
    lib.orc_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_schlick.restype = C.c_double
    lib.orc_schlick.argtypes = [_dp, _dp, C.c_double]
    lib.orc_refract.argtypes = [_dp, _dp, C.c_double, _dp]
    lib.orc_reflect.argtypes = [_dp, _dp, _dp]
    lib.orc_sky_coords.argtypes = [_dp, _dp]
    lib.orc_environment.argtypes = [C.POINTER(RtSceneDesc), _dp, _dp]
    lib.orc_camera_ray.argtypes = [C.POINTER(RtSceneDesc), C.c_double, C.c_double, _dp, C.c_int, _dp, _dp]
    lib.orc_intersect_batch.argtypes = [C.POINTER(RtSceneDesc), C.c_int64, _dp, _dp, _ip, _ip, _ip, _ip, _dp, _dp, _dp, _ip]
    lib.orc_set_origin_rule.argtypes = [C.c_int]
    lib.orc_trace_pixel.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                    C.c_uint64, _dp, C.c_int, _dp, C.POINTER(C.c_uint64), _ip,
                                    _dp, _dp, _ip, _ip, _ip, _dp, _ip]
    lib.orc_render.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_uint64, _dp, C.POINTER(C.c_uint64), C.c_int]
    lib.orc_raytrace_batch.argtypes = [C.POINTER(RtSceneDesc), C.c_int64, _dp, _dp, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                       C.c_uint64, C.c_uint32, C.c_uint32, _dp]
    lib.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.orc_counters.argtypes = [C.POINTER(C.c_uint64)]
    lib.orc_max_threads.restype = C.c_int
    return lib


@lru_cache(maxsize=None)
def skybox() -> np.ndarray:
    return np.ascontiguousarray(np.load(os.path.join(GOLD, "skybox1_rgb8.npz"))["rgb"])


@lru_cache(maxsize=None)
def packed_scene(name: str) -> PackedScene:
    """'room' or 'random_spheres_seed7' — tables packed from the upstream scene objects."""
    return PackedScene.load(os.path.join(GOLD, f"scene_{name}.npz"), sky_rgb=skybox())


def kat() -> dict:
    with open(os.path.join(GOLD, "kat.json")) as fh:
        return json.load(fh)


def load_npz(name: str):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    meta = json.loads(str(z["meta"])) if "meta" in z.files else {}
    return z, meta


class origin_rule:
    """Context manager: the oracle restates the kernels' ray-origin rule instead of upstream's
    `depth > 1e-10` self-intersection filter (see oracle/rt_oracle.c)."""

    def __enter__(self):
        oracle().orc_set_origin_rule(1)

    def __exit__(self, *a):
        oracle().orc_set_origin_rule(0)


class linear_tables:
    """Context manager: rt_set_scene keeps up to 4096 triangles in the linear tables (the default
    leaves them at a few dozen), for tests that compare the two search structures."""

    def __enter__(self):
        self.old = os.environ.get("RAYTRACING_AMD_LINEAR_MAX_TRIS")
        os.environ["RAYTRACING_AMD_LINEAR_MAX_TRIS"] = "4096"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RAYTRACING_AMD_LINEAR_MAX_TRIS", None)
        else:
            os.environ["RAYTRACING_AMD_LINEAR_MAX_TRIS"] = self.old


def oracle_intersect(scene: PackedScene, org: np.ndarray, dirs: np.ndarray, origin_prim=None):
    lib = oracle()
    n = org.shape[0]
    org = np.ascontiguousarray(org, np.float64)
    dirs = np.ascontiguousarray(dirs, np.float64)
    obj = np.empty(n, np.int32)
    prim = np.empty(n, np.int32)
    gprim = np.empty(n, np.int32)
    t = np.empty(n, np.float64)
    nrm = np.empty((n, 3), np.float64)
    pos = np.empty((n, 3), np.float64)
    nh = np.empty(n, np.int32)
    d = scene.desc()
    op_arr = None if origin_prim is None else np.ascontiguousarray(origin_prim, np.int32)
    op = None if op_arr is None else _i(op_arr)
    lib.orc_intersect_batch(C.byref(d), n, _d(org), _d(dirs), op, _i(obj), _i(prim), _i(gprim), _d(t), _d(nrm), _d(pos), _i(nh))
    return dict(obj=obj, prim=prim, gprim=gprim, t=t, normal=nrm, pos=pos, n_hits=nh)


def oracle_render(scene: PackedScene, W, H, spp, bounces, seed=SEED_RENDER, rect=None, sample_begin=0, nthreads=0):
    lib = oracle()
    arr = np.zeros((W, H, 3), np.float64)
    rays = C.c_uint64(0)
    x0, x1, y0, y1 = rect or (0, W, 0, H)
    d = scene.desc()
    lib.orc_render(C.byref(d), W, H, x0, x1, y0, y1, sample_begin, spp, bounces, seed, _d(arr), C.byref(rays), nthreads)
    return arr, int(rays.value)


def oracle_raytrace(scene: PackedScene, org, dirs, bounces, seed=SEED_RENDER, inside=None, first_depth=0, sample=0,
                    stream_begin=0) -> np.ndarray:
    """Renderer.raytrace for given rays (oracle side of HipBackend.raytrace)."""
    org = np.ascontiguousarray(org, np.float64).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, np.float64).reshape(-1, 3)
    n = org.shape[0]
    ins = None if inside is None else np.ascontiguousarray(inside, np.uint8)
    rgb = np.zeros((n, 3))
    d = scene.desc()
    oracle().orc_raytrace_batch(C.byref(d), n, _d(org), _d(dirs), None if ins is None else ins.ctypes.data_as(C.POINTER(C.c_uint8)),
                                first_depth, bounces, seed, sample, stream_begin, _d(rgb))
    return rgb


def oracle_trace_pixel(scene: PackedScene, W, H, x, y, sample, bounces, seed=SEED_RENDER, with_segments=False):
    lib = oracle()
    rgb = np.zeros(3)
    rays = C.c_uint64(0)
    nd = C.c_int32(0)
    d = scene.desc()
    if with_segments:
        cap = bounces + 2
        so, sd = np.zeros((cap, 3)), np.zeros((cap, 3))
        si, sobj, sprim = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        st = np.zeros(cap)
        ns = C.c_int32(0)
        lib.orc_trace_pixel(C.byref(d), W, H, x, y, sample, bounces, seed, None, 0, _d(rgb), C.byref(rays), C.byref(nd),
                            _d(so), _d(sd), _i(si), _i(sobj), _i(sprim), _d(st), C.byref(ns))
        k = ns.value
        return rgb, int(rays.value), nd.value, dict(org=so[:k], dir=sd[:k], inside=si[:k], obj=sobj[:k], prim=sprim[:k], t=st[:k])
    lib.orc_trace_pixel(C.byref(d), W, H, x, y, sample, bounces, seed, None, 0, _d(rgb), C.byref(rays), C.byref(nd),
                        None, None, None, None, None, None, None)
    return rgb, int(rays.value), nd.value, None
