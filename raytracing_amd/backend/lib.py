"""ctypes binding of the C ABI in include/raytracing_amd.h (``_rt_hip.so``, built by
``python -m raytracing_amd.backend.build`` or ``__graft_entry__.build()``).

There is no CPU implementation behind these calls: if the shared library is missing, or no
gfx950 device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .packer import PackedScene, RtSceneDesc, RT_SCENE_REFIT

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_rt_hip.so")

STATUS_NAMES = {0: "RT_OK", 1: "RT_ERR_INVALID_ARG", 2: "RT_ERR_NO_SCENE", 3: "RT_ERR_HIP", 4: "RT_ERR_NO_DEVICE",
                5: "RT_ERR_STOPPED", 6: "RT_ERR_UNSUPPORTED"}
RT_ERR_STOPPED = 5

# every symbol include/raytracing_amd.h declares
EXPORTS = ["rt_abi_version", "rt_create", "rt_destroy", "rt_last_error", "rt_set_scene", "rt_intersect_batch",
           "rt_accum_resize", "rt_accum_clear", "rt_accum_device_ptr", "rt_accum_read", "rt_render", "rt_synchronize",
           "rt_comm_available", "rt_comm_unique_id", "rt_comm_init", "rt_merge",
           "rt_release_scratch", "rt_request_stop", "rt_clear_stop", "rt_get_stats", "rt_reset_stats", "rt_debug_uniforms", "rt_debug_scene_layout", "rt_debug_walk_fallbacks", "rt_debug_render_counters",
           "rt_debug_trace_paths", "rt_raytrace_batch", "rt_postprocess", "rt_postprocess_noalloc", "rt_build_bvh", "rt_collapse_bvh", "rt_build_bvh_threads", "rt_obj_open", "rt_obj_read", "rt_obj_close"]


class RtRenderParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
                ("x0", C.c_int32), ("x1", C.c_int32), ("y0", C.c_int32), ("y1", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_count", C.c_int32), ("max_bounces", C.c_int32),
                ("seed", C.c_uint64), ("flags", C.c_uint32), ("wavefront_paths", C.c_int32)]


RT_RENDER_WAVEFRONT, RT_RENDER_STRICT_DEPTH, RT_RENDER_NO_DEFER, RT_RENDER_LOOSE_DEPTH = 1, 2, 4, 8


class RtPostprocessParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("exposure", C.c_float), ("brightness", C.c_float), ("contrast", C.c_float),
                ("saturation", C.c_float), ("gamma", C.c_float), ("key", C.c_float), ("tone_saturation", C.c_float)]


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("sphere_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("aabb_tests", C.c_uint64), ("bvh_node_tests", C.c_uint64), ("exact_rays", C.c_uint64),
                ("refined_hits", C.c_uint64), ("sky_lookups", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("kernel_ms", C.c_double), ("launches", C.c_uint64), ("kernel_ms_min", C.c_double), ("kernel_ms_max", C.c_double)]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class BackendError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load ``_rt_hip.so`` and declare the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RAYTRACING_AMD_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build the HIP library first (python -m raytracing_amd.backend.build). "
                           "This package has no CPU render path.")
    lib = C.CDLL(p)
    vp, dp, ip, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    lib.rt_abi_version.restype = C.c_int
    lib.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_set_scene.argtypes = [vp, C.POINTER(RtSceneDesc)]
    lib.rt_intersect_batch.argtypes = [vp, C.c_int64, dp, dp, ip, ip, ip, ip, ip, dp, dp, dp]
    lib.rt_accum_resize.argtypes = [vp, C.c_int32, C.c_int32]
    lib.rt_accum_clear.argtypes = [vp]
    lib.rt_accum_device_ptr.argtypes = [vp]
    lib.rt_accum_device_ptr.restype = vp
    lib.rt_accum_read.argtypes = [vp, fp]
    lib.rt_render.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp]
    lib.rt_synchronize.argtypes = [vp, vp]
    lib.rt_comm_unique_id.argtypes = [vp]
    lib.rt_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.rt_merge.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    lib.rt_request_stop.argtypes = [vp]
    lib.rt_release_scratch.argtypes = [vp]
    lib.rt_clear_stop.argtypes = [vp]
    lib.rt_get_stats.argtypes = [vp, C.POINTER(RtStats)]
    lib.rt_reset_stats.argtypes = [vp]
    lib.rt_debug_uniforms.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int32, fp]
    lib.rt_debug_scene_layout.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, fp]
    lib.rt_debug_render_counters.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.rt_debug_walk_fallbacks.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.rt_debug_trace_paths.argtypes = [vp, C.c_int32, ip, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32,
                                         fp, ip, ip, ip, ip, fp, fp, fp, fp]
    lib.rt_raytrace_batch.argtypes = [vp, C.c_int64, dp, dp, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_uint64,
                                      C.c_uint32, C.c_uint32, fp]
    lib.rt_postprocess.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.POINTER(RtPostprocessParams), vp, vp]
    lib.rt_postprocess_noalloc.argtypes = lib.rt_postprocess.argtypes
    lib.rt_build_bvh_threads.argtypes = [C.c_int32]
    lib.rt_collapse_bvh.argtypes = [C.c_int32, vp, vp, C.c_int32, ip, ip]
    lib.rt_build_bvh.argtypes = [C.c_int32, dp, C.c_int32, vp, C.c_int32, ip, ip]
    lp = C.POINTER(C.c_int64)
    lib.rt_obj_open.argtypes = [C.c_char_p, C.POINTER(vp), ip, lp, lp]
    lib.rt_obj_read.argtypes = [vp, C.c_char_p, C.c_int32, lp, lp, dp, ip]
    lib.rt_obj_close.argtypes = [vp]
    if path is None:
        _lib = lib
    return lib


class HipBackend:
    """One context on one GPU: scene upload, closest-hit queries, rendering into a device buffer."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.rt_create(device, C.byref(h))
        if st != 0:
            raise BackendError(st, (self.lib.rt_last_error(None) or b"").decode())
        self.handle = h
        self.device = device
        self._scene_keepalive = None
        self.size = None
        self.comm_ranks = 0   # > 0 once comm_init has run

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, st: int) -> None:
        if st != 0:
            raise BackendError(st, (self.lib.rt_last_error(self.handle) or b"").decode())

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ----------------------------------------------------------------------------------
    def set_scene(self, packed: PackedScene, refit: bool = False) -> None:
        """Upload a scene.  ``refit``: the scene differs from the current one only in vertex positions
        (moved meshes): keep the hierarchy's topology and refit its boxes (``RT_SCENE_REFIT``; the
        library checks that the meshes match and builds anew when they do not)."""
        d = packed.desc()
        if refit:
            d.flags |= RT_SCENE_REFIT
        self._check(self.lib.rt_set_scene(self.handle, C.byref(d)))
        self._scene_keepalive = packed

    # -- closest hit ----------------------------------------------------------------------------
    def intersect(self, origins, directions, origin_prim=None) -> dict:
        org = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        drs = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = org.shape[0]
        if drs.shape[0] != n:
            raise ValueError("origins and directions differ in length")
        out = dict(obj=np.empty(n, np.int32), prim=np.empty(n, np.int32), gprim=np.empty(n, np.int32),
                   how=np.empty(n, np.int32), t=np.empty(n, np.float64), normal=np.empty((n, 3), np.float64),
                   pos=np.empty((n, 3), np.float64))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        op = None
        if origin_prim is not None:
            origin_prim = np.ascontiguousarray(origin_prim, dtype=np.int32)
            op = origin_prim.ctypes.data_as(ip)
        self._check(self.lib.rt_intersect_batch(
            self.handle, n, org.ctypes.data_as(dp), drs.ctypes.data_as(dp), op,
            out["obj"].ctypes.data_as(ip), out["prim"].ctypes.data_as(ip), out["gprim"].ctypes.data_as(ip),
            out["how"].ctypes.data_as(ip), out["t"].ctypes.data_as(dp), out["normal"].ctypes.data_as(dp),
            out["pos"].ctypes.data_as(dp)))
        return out

    # -- accumulation buffer --------------------------------------------------------------------
    def resize(self, width: int, height: int) -> None:
        self._check(self.lib.rt_accum_resize(self.handle, width, height))
        self.size = (width, height)

    def clear(self) -> None:
        self._check(self.lib.rt_accum_clear(self.handle))

    def accum_ptr(self) -> int:
        return int(self.lib.rt_accum_device_ptr(self.handle) or 0)

    def read(self) -> np.ndarray:
        w, h = self.size
        out = np.empty((w, h, 3), dtype=np.float32)
        self._check(self.lib.rt_accum_read(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- render ---------------------------------------------------------------------------------
    def render(self, width: int, height: int, sample_begin: int, sample_count: int, max_bounces: int, seed: int,
               rect=None, accum_ptr: int = 0, stream: int = 0, strict_depth: bool = True, wavefront=None,
               wavefront_paths: int = 0, no_defer=None) -> None:
        """``wavefront`` None takes the switch from the environment (RAYTRACING_AMD_WAVEFRONT /
        RAYTRACING_AMD_WF_PATHS: A/B runs and tests); the library itself reads no environment variable.
        ``no_defer`` (None: RAYTRACING_AMD_NO_DEFER): the one full kernel instead of the lean kernel
        plus the set-aside paths (``RT_RENDER_NO_DEFER``).  ``strict_depth=False`` asks for the looser depth
        acceptance of earlier versions (``RT_RENDER_LOOSE_DEPTH``, A/B runs only)."""
        x0, x1, y0, y1 = rect if rect is not None else (0, width, 0, height)
        if wavefront is None:
            wavefront = bool(os.environ.get("RAYTRACING_AMD_WAVEFRONT"))
            try:
                wavefront_paths = wavefront_paths or int(os.environ.get("RAYTRACING_AMD_WF_PATHS", "0") or 0)
            except ValueError:
                wavefront_paths = 0
        if no_defer is None:
            no_defer = bool(os.environ.get("RAYTRACING_AMD_NO_DEFER"))
        flags = (RT_RENDER_WAVEFRONT if wavefront else 0) | (0 if strict_depth else RT_RENDER_LOOSE_DEPTH) | (RT_RENDER_NO_DEFER if no_defer else 0)
        p = RtRenderParams(C.sizeof(RtRenderParams), width, height, x0, x1, y0, y1, sample_begin, sample_count,
                           max_bounces, seed & 0xFFFFFFFFFFFFFFFF, flags, max(0, int(wavefront_paths)))
        self._check(self.lib.rt_render(self.handle, C.byref(p), C.c_void_p(accum_ptr or None), C.c_void_p(stream or None)))

    def postprocess(self, width: int, height: int, dst_ptr: int, src_ptr: int = 0, scale: float = 1.0, flip_y: bool = False,
                    exposure=1.0, brightness=0.0, contrast=1.0, saturation=1.0, gamma=1.0, key=0.18, tone_saturation=1.0,
                    stream: int = 0, no_alloc: bool = False) -> None:
        """Fused post-processing chain from a (W, H, 3) float32 device buffer (0 = the context's
        accumulation buffer) into another; asynchronous on ``stream``.  ``no_alloc`` takes the entry
        point that works over the library's fixed scratch array and allocates nothing."""
        p = RtPostprocessParams(C.sizeof(RtPostprocessParams), exposure, brightness, contrast, saturation, gamma, key,
                                tone_saturation)
        entry = self.lib.rt_postprocess_noalloc if no_alloc else self.lib.rt_postprocess
        self._check(entry(self.handle, C.c_void_p(src_ptr or None), width, height, scale, int(bool(flip_y)),
                          C.byref(p), C.c_void_p(dst_ptr), C.c_void_p(stream or None)))

    # -- merge across GPUs (RCCL, issued by the library) ------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte communicator id: rank 0 makes it and hands it to every other rank."""
        lib = load_library()
        buf = C.create_string_buffer(128)
        st = lib.rt_comm_unique_id(buf)
        if st != 0:
            raise BackendError(st, (lib.rt_last_error(None) or b"").decode())
        return buf.raw

    @staticmethod
    def comm_ready() -> bool:
        """Whether this process can open RCCL at all (``rt_comm_available``): asked on every rank before the
        collective ``comm_init``, so that no rank enters it alone."""
        return load_library().rt_comm_available() == 0

    def comm_init(self, n_ranks: int, rank: int, unique_id: Optional[bytes] = None) -> None:
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.rt_comm_init(self.handle, n_ranks, rank, buf))
        self.comm_ranks = n_ranks

    def merge(self, width: int, height: int, accum_ptr: int = 0, stream: int = 0) -> None:
        """In-place sum of the (W, H, 3) float32 buffers of all ranks (0 = the context's own buffer)."""
        self._check(self.lib.rt_merge(self.handle, C.c_void_p(accum_ptr or None), width, height, C.c_void_p(stream or None)))

    def synchronize(self, stream: int = 0) -> None:
        self._check(self.lib.rt_synchronize(self.handle, C.c_void_p(stream or None)))

    def request_stop(self) -> None:
        self.lib.rt_request_stop(self.handle)

    def clear_stop(self) -> None:
        self.lib.rt_clear_stop(self.handle)

    def stats(self) -> dict:
        s = RtStats()
        self._check(self.lib.rt_get_stats(self.handle, C.byref(s)))
        return s.as_dict()

    def reset_stats(self) -> None:
        self._check(self.lib.rt_reset_stats(self.handle))

    def debug_uniforms(self, pixel: int, sample: int, seed: int, n_segments: int) -> np.ndarray:
        out = np.empty((n_segments, 5), dtype=np.float32)
        self._check(self.lib.rt_debug_uniforms(self.handle, pixel, sample, seed, n_segments,
                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def release_scratch(self) -> None:
        """Give the context's grow-only scratch memory back to the device (``rt_release_scratch``)."""
        self._check(self.lib.rt_release_scratch(self.handle))

    def render_counters(self) -> dict:
        """What the last ``render`` call set aside (tests and tuning; synchronises).  ``jobs`` and
        ``jobs_rendered_again`` are hand-out counters: each wave overshoots them by its last window."""
        out = (C.c_uint32 * 4)()
        self._check(self.lib.rt_debug_render_counters(self.handle, out))
        return dict(jobs=out[0], paths_set_aside=out[1], jobs_given_up=out[2], jobs_rendered_again=out[3])

    def scene_layout(self) -> dict:
        """How ``rt_set_scene`` laid the current scene's triangles out (tests and tuning)."""
        out = (C.c_int32 * 10)()
        slack = C.c_float(0.0)
        self._check(self.lib.rt_debug_scene_layout(self.handle, out, 10, C.byref(slack)))
        return dict(quad_rows=out[0], quad_general=out[1], compact_rows=(out[2], out[3], out[4]), tri_rows=out[5],
                    boxes=out[6], bvh_nodes=out[7], refitted=bool(out[8]), walk_stack_need=out[9], box_slack=float(slack.value))

    def walk_fallbacks(self) -> int:
        """Walks of the hierarchy that ran out of stack and took the stackless walk of last resort (since the library was loaded)."""
        out = C.c_uint64(0)
        self._check(self.lib.rt_debug_walk_fallbacks(self.handle, C.byref(out)))
        return int(out.value)

    def raytrace(self, origins, directions, max_bounces: int, seed: int = 0, inside=None, first_depth: int = 0,
                 sample: int = 0, stream_begin: int = 0) -> np.ndarray:
        """Radiance of the paths that start with the given rays (upstream's ``Renderer.raytrace``)."""
        org = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        drs = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = org.shape[0]
        if drs.shape[0] != n:
            raise ValueError("origins and directions differ in length")
        ins = None
        if inside is not None:
            ins_arr = np.ascontiguousarray(inside, dtype=np.uint8).reshape(-1)
            if ins_arr.shape[0] != n:
                raise ValueError("inside flags and rays differ in length")
            ins = ins_arr.ctypes.data_as(C.POINTER(C.c_uint8))
        out = np.empty((n, 3), dtype=np.float32)
        dp = C.POINTER(C.c_double)
        self._check(self.lib.rt_raytrace_batch(self.handle, n, org.ctypes.data_as(dp), drs.ctypes.data_as(dp), ins,
                                               first_depth, max_bounces, seed & 0xFFFFFFFFFFFFFFFF, sample, stream_begin,
                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def trace_paths(self, xys, width: int, height: int, max_bounces: int, seed: int, strict_depth: bool = True) -> dict:
        """Whole paths with per-segment records (diagnostics / parity tests).  ``org`` is the origin each segment's
        ray starts from as the exact code sees it, in double precision: the float32 origin (``org32``) plus, behind a hit
        the double-precision code produced, the part of that hit's position float32 cannot hold."""
        xys = np.ascontiguousarray(xys, dtype=np.int32).reshape(-1, 3)
        n, cap = xys.shape[0], max_bounces + 1
        out = dict(rgb=np.empty((n, 3), np.float32), n_seg=np.empty(n, np.int32), obj=np.empty((n, cap), np.int32),
                   prim=np.empty((n, cap), np.int32), how=np.empty((n, cap), np.int32), t=np.empty((n, cap), np.float32),
                   org32=np.empty((n, cap, 3), np.float32), dir=np.empty((n, cap, 3), np.float32), org_res=np.empty((n, cap, 3), np.float32))
        ip, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        self._check(self.lib.rt_debug_trace_paths(
            self.handle, n, xys.ctypes.data_as(ip), width, height, max_bounces, seed & 0xFFFFFFFFFFFFFFFF,
            0 if strict_depth else RT_RENDER_LOOSE_DEPTH,
            out["rgb"].ctypes.data_as(fp), out["n_seg"].ctypes.data_as(ip), out["obj"].ctypes.data_as(ip),
            out["prim"].ctypes.data_as(ip), out["how"].ctypes.data_as(ip), out["t"].ctypes.data_as(fp),
            out["org32"].ctypes.data_as(fp), out["dir"].ctypes.data_as(fp), out["org_res"].ctypes.data_as(fp)))
        out["org"] = out["org32"].astype(np.float64) + out["org_res"].astype(np.float64)
        return out
