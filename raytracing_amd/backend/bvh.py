"""Flat bounding-volume hierarchy over a scene's triangles, built on the host by the native
builder in the HIP library (``rt_build_bvh``: binned surface-area heuristic, C++; it runs on the
CPU and needs no GPU).

Upstream has no acceleration structure (src/mesh/mesh.py:94-107 tests every face for every ray);
the hierarchy replaces that loop and must return the same hits, which ``rt_set_scene`` enforces by
checking coverage and containment of whatever hierarchy it is handed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

#: numpy view of ``rt_bvh_node`` (include/raytracing_amd.h)
NODE_DTYPE = np.dtype([("lo", np.float32, (3,)), ("skip", np.int32), ("hi", np.float32, (3,)), ("tri", np.int32)])
assert NODE_DTYPE.itemsize == 32


def build_flat_bvh(tri_verts: np.ndarray, leaf_size: int = 3) -> tuple[np.ndarray, np.ndarray]:
    """(nodes, tri_index) for ``tri_verts`` of shape (n, 9) float64.

    ``nodes`` is a structured array in threaded depth-first order: the first child of an inner node
    is the next node, ``skip`` is where a walk continues after the node's subtree (-1 = done);
    leaves hold ``tri = (first << 4) | count`` into ``tri_index``, inner nodes ``tri = -1``.
    """
    from .lib import load_library

    tv = np.ascontiguousarray(tri_verts, dtype=np.float64).reshape(-1, 9)
    n = int(tv.shape[0])
    if n == 0:
        return np.zeros(0, NODE_DTYPE), np.zeros(0, np.int32)
    lib = load_library()
    from .packer import layout_switches

    lib.rt_build_bvh_threads(layout_switches()["build_threads"])   # RAYTRACING_AMD_BUILD_THREADS (0 = hardware threads)
    nodes = np.zeros(2 * n, NODE_DTYPE)
    order = np.zeros(n, np.int32)
    n_nodes = C.c_int32(0)
    st = lib.rt_build_bvh(n, tv.ctypes.data_as(C.POINTER(C.c_double)), int(leaf_size), nodes.ctypes.data_as(C.c_void_p),
                          int(nodes.shape[0]), order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_nodes))
    if st != 0:
        raise RuntimeError(f"rt_build_bvh failed with status {st}")
    return np.ascontiguousarray(nodes[: n_nodes.value]), order


def check_flat_bvh(nodes: np.ndarray, tri_index: np.ndarray, tri_verts: np.ndarray) -> None:
    """Independent (numpy) restatement of the library's acceptance checks; raises ValueError."""
    n_nodes, n_tris = len(nodes), len(tri_index)
    tv = np.asarray(tri_verts, np.float64).reshape(-1, 3, 3)
    skip = np.where(nodes["skip"] < 0, n_nodes, nodes["skip"])
    idx = np.arange(n_nodes)
    if not np.all(skip > idx) or not np.all(skip <= n_nodes):
        raise ValueError("skip links must point forward")
    leaf = nodes["tri"] >= 0
    first, cnt = nodes["tri"] >> 4, nodes["tri"] & 15
    if np.any(leaf & ((cnt < 1) | (first < 0) | (first + cnt > n_tris))):
        raise ValueError("bad leaf range")
    if np.any(leaf & (skip != idx + 1)) or np.any(~leaf & (skip == idx + 1)):
        raise ValueError("leaf / inner node shape")
    covered = np.zeros(n_tris, np.int64)
    for i in np.nonzero(leaf)[0]:
        ids = tri_index[first[i]: first[i] + cnt[i]]
        np.add.at(covered, ids, 1)
        v = tv[ids].reshape(-1, 3)
        if np.any(v < nodes["lo"][i].astype(np.float64)) or np.any(v > nodes["hi"][i].astype(np.float64)):
            raise ValueError(f"leaf {i}: triangle outside its box")
    if not np.all(covered == 1):
        raise ValueError("every triangle must be in exactly one leaf")
    # containment in the parent: walk with a stack of open subtrees
    stack: list[tuple[int, int]] = []
    for i in range(n_nodes):
        while stack and stack[-1][0] <= i:
            stack.pop()
        if stack:
            p = stack[-1][1]
            if skip[i] > stack[-1][0] or np.any(nodes["lo"][i] < nodes["lo"][p]) or np.any(nodes["hi"][i] > nodes["hi"][p]):
                raise ValueError(f"node {i} not inside its parent {p}")
        elif i != 0:
            raise ValueError(f"node {i} unreachable")
        if not leaf[i]:
            stack.append((int(skip[i]), i))


#: numpy view of the four-wide node the kernels walk (``rt_collapse_bvh`` in include/raytracing_amd.h)
WIDE_DTYPE = np.dtype([("cx", np.float32, (4,)), ("hx", np.float32, (4,)), ("cy", np.float32, (4,)), ("hy", np.float32, (4,)),
                       ("cz", np.float32, (4,)), ("hz", np.float32, (4,)), ("link", np.int32, (4,)), ("order", np.uint32, (2,)),
                       ("first", np.int32), ("inner", np.uint32)])
assert WIDE_DTYPE.itemsize == 128


def collapse_flat_bvh(nodes: np.ndarray) -> tuple[np.ndarray, int]:
    """(wide nodes, stack entries the deepest walk can need) for a hierarchy in ``build_flat_bvh``'s layout: the
    four-wide form ``rt_set_scene`` makes for the kernels (CPU only)."""
    from .lib import load_library

    lib = load_library()
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    wide = np.zeros(max(len(nodes), 1), WIDE_DTYPE)
    n_wide, need = C.c_int32(0), C.c_int32(0)
    st = lib.rt_collapse_bvh(len(nodes), nodes.ctypes.data_as(C.c_void_p), wide.ctypes.data_as(C.c_void_p), len(wide),
                             C.byref(n_wide), C.byref(need))
    if st != 0:
        raise RuntimeError(f"rt_collapse_bvh failed with status {st}")
    return np.ascontiguousarray(wide[: n_wide.value]), int(need.value)
