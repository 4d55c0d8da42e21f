"""Scene objects: ``ObjectBase``, ``Sphere``, ``Mesh`` (+ Wavefront OBJ loading).

Mirrors the upstream object model (src/mesh/base.py:8-22, src/mesh/sphere.py:8-11,
src/mesh/mesh.py:9-59,145-251) as far as SCENE DESCRIPTION goes: the constructor signatures, the
attribute names the packer reads (``material``, ``origin``, ``radius``, ``transformed_vertices``,
``triangles``, ``bounding_box``, ``name``) and the transform order scale -> rotate X,Y,Z (degrees)
-> translate.  Vertex data is held in numpy arrays instead of lists of vectors so that meshes with
millions of faces load and pack in seconds.

Ray intersection does not live here: upstream's per-object ``intersect`` loops
(sphere.py:13-41, mesh.py:61-143) are what the HIP kernels replace.  ``obj.intersect(ray)`` is kept
as a convenience that asks the GPU backend about a one-object scene.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

from .core import HitInfo, Material, Ray, Vec3

__all__ = ["ObjectBase", "Sphere", "Mesh"]


class ObjectBase:
    def __init__(self, material: Material, origin: Vec3) -> None:
        self.material = material
        self.origin = origin
        self.animations: list = []
        self.id: int = id(self)
        self.name: str = "Object"

    def update(self, t) -> None:
        for animation in self.animations:
            animation.update(t)

    def intersect(self, ray: Ray) -> List[HitInfo]:
        """Closest hit of ``ray`` with this object alone, computed on the GPU (0 or 1 records)."""
        if type(self) is ObjectBase:
            raise NotImplementedError()
        from .raytracing import Scene  # local import: raytracing imports this module

        probe = Scene()
        probe.add_object(self)
        return probe.intersect(ray)


class Sphere(ObjectBase):
    def __init__(self, material: Material, origin: Vec3, radius: float) -> None:
        super().__init__(material, origin)
        self.radius: float = radius


def _rotation_matrix(rotation_deg: Sequence[float]) -> np.ndarray:
    """Matrix of: rotate about +X by rx, then about +Y by ry, then about +Z by rz (degrees,
    right-handed), i.e. v' = Rz @ Ry @ Rx @ v (mesh.py:151-161)."""
    rx, ry, rz = (math.radians(float(a)) for a in rotation_deg)
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    mx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    my = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    mz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return mz @ my @ mx


class Mesh(ObjectBase):
    """Indexed triangle mesh with scale / rotation / translation.

    ``vertices`` may be any (n,3) array-like (a list of ``Vec3`` works); ``triangles`` any (m,3)
    integer array-like indexing ``vertices``.
    """

    _warned_no_native_reader = False   # the fallback to the Python OBJ parser is announced once per process

    def __init__(self, material: Material, origin: Vec3, vertices, triangles,
                 scale: Vec3 | None = None, rotation: Vec3 | None = None) -> None:
        super().__init__(material, origin)
        # Change counter: every assignment to vertices / triangles / transformed_vertices bumps it, so
        # that a scene can tell "this mesh changed since it was uploaded" without reading a vertex and
        # without relying on array identities (CPython reuses the addresses of freed arrays).
        self._version = 0
        self._vertices = np.zeros((0, 3))
        self._triangles = np.zeros((0, 3), dtype=np.int32)
        self.scale: Vec3 = scale if scale else Vec3(1)
        self.rotation: Vec3 = rotation if rotation else Vec3(0)
        self.transformed_vertices = np.zeros((0, 3))
        self.bounding_box: tuple[Vec3, Vec3] | None = None
        self.vertices = vertices
        self.triangles = triangles
        self.update_transformed_vertices()

    # vertices / triangles are normalised to arrays on assignment
    @property
    def vertices(self) -> np.ndarray:
        return self._vertices

    @vertices.setter
    def vertices(self, value) -> None:
        arr = np.asarray([tuple(v) for v in value] if len(value) and not isinstance(value, np.ndarray) else value,
                         dtype=np.float64)
        self._vertices = arr.reshape(-1, 3)
        self._version += 1

    @property
    def triangles(self) -> np.ndarray:
        return self._triangles

    @triangles.setter
    def triangles(self, value) -> None:
        self._triangles = np.asarray(value, dtype=np.int32).reshape(-1, 3)
        self._version += 1

    @property
    def transformed_vertices(self) -> np.ndarray:
        return self._transformed_vertices

    @transformed_vertices.setter
    def transformed_vertices(self, value) -> None:
        self._transformed_vertices = np.asarray(value, dtype=np.float64).reshape(-1, 3)
        self._version += 1

    @property
    def version(self) -> int:
        """Bumped by every change of the geometry (setters, transforms, ``touch``)."""
        return self._version

    def touch(self) -> None:
        """Declare an in-place edit of one of the arrays (``mesh.transformed_vertices[i] = ...``)."""
        self._version += 1

    # -- transforms (mesh.py:145-176) ----------------------------------------------------------
    def get_scaled_vertices(self, points) -> np.ndarray:
        return np.asarray(points, dtype=np.float64).reshape(-1, 3) * np.asarray(tuple(self.scale))

    def get_rotated_vertices(self, vertices) -> np.ndarray:
        v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        if tuple(self.rotation) == (0.0, 0.0, 0.0):
            return v.copy()
        return v @ _rotation_matrix(tuple(self.rotation)).T

    def get_translated_vertices(self, points) -> np.ndarray:
        return np.asarray(points, dtype=np.float64).reshape(-1, 3) + np.asarray(tuple(self.origin))

    def get_transformed_vertices(self) -> np.ndarray:
        return self.get_translated_vertices(self.get_rotated_vertices(self.get_scaled_vertices(self._vertices)))

    def update_transformed_vertices(self) -> None:
        self.transformed_vertices = self.get_transformed_vertices()
        self.update_bounding_box()

    def update_bounding_box(self) -> None:
        tv = self.transformed_vertices
        if tv.shape[0] == 0:
            self.bounding_box = None
            return
        self.bounding_box = (Vec3(tv.min(axis=0)), Vec3(tv.max(axis=0)))

    # -- setters that re-derive the cached data (mesh.py:178-209) -------------------------------
    def set_scale(self, scale: Vec3) -> None:
        self.scale = scale
        self.update_transformed_vertices()

    def set_origin(self, origin: Vec3) -> None:
        self.origin = origin
        self.update_transformed_vertices()

    def set_rotation(self, rotation: Vec3) -> None:
        self.rotation = rotation
        self.update_transformed_vertices()

    def set_origin_to_center_of_mass(self) -> None:
        if self._vertices.shape[0] == 0:
            return
        com = self._vertices.sum(axis=0) / self._vertices.shape[0]
        self._vertices = self._vertices - com
        self._version += 1
        self.origin = Vec3(self.origin) + Vec3(com)
        self.update_transformed_vertices()

    # -- OBJ loading (mesh.py:211-251) ---------------------------------------------------------
    @staticmethod
    def _parse_obj_python(path: str) -> list:
        """[(name, vertices, triangles)] per ``o`` record: the plain-Python statement of the
        semantics (used when the native library is not built, and as the yardstick in the tests)."""
        groups: list = []
        verts: list = []
        tris: list = []
        offset = 0
        with open(path, "r") as fh:
            for raw in fh:
                line = raw.strip()
                tag = line[:2]
                if tag == "o ":
                    if groups:
                        groups[-1] = (groups[-1][0], verts, tris)
                        offset += len(verts)
                        verts, tris = [], []
                    groups.append((line.split(" ", 1)[1], [], []))
                elif tag == "v ":
                    parts = line.split()
                    verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
                elif tag == "f ":
                    idx = [int(tok.split("/")[0]) - 1 - offset for tok in line.split()[1:]]
                    for k in range(1, len(idx) - 1):
                        tris.append((idx[0], idx[k], idx[k + 1]))
        if groups:
            groups[-1] = (groups[-1][0], verts, tris)
        return groups

    @staticmethod
    def _parse_obj_native(path: str):
        """Same result from the C++ reader in the HIP library (rt_obj_open / rt_obj_read: host code,
        needs no GPU); None when the library is not built."""
        import ctypes as C

        try:
            from .backend.lib import load_library

            lib = load_library()
        except Exception as e:   # noqa: BLE001 - the pure-Python reader gives the same result, some thirty times slower
            if not Mesh._warned_no_native_reader:
                import warnings

                Mesh._warned_no_native_reader = True
                warnings.warn(f"raytracing_amd: the native OBJ reader is unavailable ({e!r}); falling back to the "
                              "Python parser, which is about 30 times slower on large files", RuntimeWarning, stacklevel=3)
            return None
        handle = C.c_void_p()
        n_obj, n_v, n_t = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        st = lib.rt_obj_open(str(path).encode(), C.byref(handle), C.byref(n_obj), C.byref(n_v), C.byref(n_t))
        if st != 0:
            raise ValueError((lib.rt_last_error(None) or b"").decode() or f"cannot read {path}")
        try:
            n = n_obj.value
            stride = 256
            names = C.create_string_buffer(max(n, 1) * stride)
            vb = np.zeros(n + 1, np.int64)
            tb = np.zeros(n + 1, np.int64)
            v = np.zeros((n_v.value, 3), np.float64)
            t = np.zeros((n_t.value, 3), np.int32)
            lib.rt_obj_read(handle, names, stride, vb.ctypes.data_as(C.POINTER(C.c_int64)), tb.ctypes.data_as(C.POINTER(C.c_int64)),
                            v.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_int32)))
        finally:
            lib.rt_obj_close(handle)
        return [(names.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode(), v[vb[i]:vb[i + 1]], t[tb[i]:tb[i + 1]])
                for i in range(n)]

    @classmethod
    def load_from_obj_file(cls, path: str) -> List["Mesh"]:
        """One ``Mesh`` per ``o`` record, in file order.

        Semantics kept from upstream because they define object and face numbering: only ``o``,
        ``v`` and ``f`` records are read; a face corner ``a/b/c`` contributes its first index; face
        indices are rebased by the number of vertices of all earlier objects; polygons are
        fan-triangulated around their first corner; records before the first ``o`` are collected
        but dropped if no object ever starts.
        """
        groups = cls._parse_obj_native(path)
        if groups is None:
            groups = cls._parse_obj_python(path)
        meshes: List[Mesh] = []
        for name, v, t in groups:
            m = cls(Material.default_material(), Vec3(0), np.asarray(v, dtype=np.float64).reshape(-1, 3),
                    np.asarray(t, dtype=np.int32).reshape(-1, 3))
            m.name = name
            meshes.append(m)
        return meshes
