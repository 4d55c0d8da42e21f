"""Scene -> flat structure-of-arrays tables (the ``rt_scene_desc`` of include/raytracing_amd.h).

The packer is duck-typed: it reads only attributes, so it accepts this package's scene objects AND
the upstream project's own objects (a maintainer can hand an upstream ``Scene`` to the GPU backend
unchanged).  Attributes read, with the upstream definitions they correspond to:

  scene.objects                         src/raytracing/scene.py:15     (order = hit tie-break order)
  obj.material.{color,emission_strength,smoothness,transmittance,ior}  src/core/material.py:17-21
  sphere: obj.origin, obj.radius        src/mesh/sphere.py:9-11
  mesh:   obj.transformed_vertices, obj.triangles, obj.bounding_box    src/mesh/mesh.py:25-59
  scene.camera.{pos,right,up,forward,d,dof_strength,dof_dist}          src/raytracing/camera.py:17-37
  environment: ``scene.skybox`` (this package) or the module-level ``sky_map`` captured by the
  function installed as ``scene.get_environment``                       scenes/room.py:10-22,48

Everything is kept in float64 exactly as the Python objects hold it; the device library derives
its float32 working set from these tables.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

OBJ_SPHERE = 0
OBJ_MESH = 1


class RtSceneDesc(C.Structure):
    """ctypes image of ``rt_scene_desc``."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_objects", C.c_int32),
        ("obj_kind", C.POINTER(C.c_int32)),
        ("obj_index", C.POINTER(C.c_int32)),
        ("materials", C.POINTER(C.c_double)),
        ("n_spheres", C.c_int32),
        ("spheres", C.POINTER(C.c_double)),
        ("sphere_obj", C.POINTER(C.c_int32)),
        ("n_meshes", C.c_int32),
        ("mesh_aabb", C.POINTER(C.c_double)),
        ("mesh_obj", C.POINTER(C.c_int32)),
        ("mesh_tri_begin", C.POINTER(C.c_int32)),
        ("n_tris", C.c_int32),
        ("tri_verts", C.POINTER(C.c_double)),
        ("cam_pos", C.c_double * 3),
        ("cam_right", C.c_double * 3),
        ("cam_up", C.c_double * 3),
        ("cam_forward", C.c_double * 3),
        ("cam_d", C.c_double),
        ("cam_dof_strength", C.c_double),
        ("cam_dof_dist", C.c_double),
        ("sky_w", C.c_int32),
        ("sky_h", C.c_int32),
        ("sky_rgb", C.POINTER(C.c_uint8)),
        ("sky_scale", C.c_double),
        ("n_bvh_nodes", C.c_int32),
        ("bvh_nodes", C.c_void_p),
        ("bvh_tri_index", C.POINTER(C.c_int32)),
        ("flags", C.c_uint32),
        ("linear_max_tris", C.c_int32),
        ("bvh_leaf_size", C.c_int32),
        ("build_threads", C.c_int32),
    ]


# rt_scene_desc.flags
RT_SCENE_NO_PAIR_ROWS, RT_SCENE_NO_COMPACT_ROWS, RT_SCENE_NO_LINEAR_SPLIT, RT_SCENE_TINY_WALK_STACK, RT_SCENE_NO_BOX_ROWS, RT_SCENE_REFIT = 1, 2, 4, 8, 16, 32


def layout_switches() -> dict:
    """The layout switches of ``rt_scene_desc`` as the environment sets them (all optional, for A/B
    runs, tests and tuning; the library itself reads no environment variable)."""
    env = os.environ
    flags = 0
    if env.get("RAYTRACING_AMD_NO_QUADS"):
        flags |= RT_SCENE_NO_PAIR_ROWS
    if env.get("RAYTRACING_AMD_NO_SLABS"):
        flags |= RT_SCENE_NO_COMPACT_ROWS
    if env.get("RAYTRACING_AMD_NO_LINEAR_SPLIT"):
        flags |= RT_SCENE_NO_LINEAR_SPLIT
    if env.get("RAYTRACING_AMD_TINY_WALK_STACK"):
        flags |= RT_SCENE_TINY_WALK_STACK
    if env.get("RAYTRACING_AMD_NO_BOXES"):
        flags |= RT_SCENE_NO_BOX_ROWS

    def num(name):
        try:
            return max(0, int(env.get(name, "0") or 0))
        except ValueError:
            return 0

    return dict(flags=flags, linear_max_tris=num("RAYTRACING_AMD_LINEAR_MAX_TRIS"), bvh_leaf_size=num("RAYTRACING_AMD_BVH_LEAF"),
                build_threads=num("RAYTRACING_AMD_BUILD_THREADS"))


def _vec(v) -> tuple[float, float, float]:
    x, y, z = v
    return float(x), float(y), float(z)


def _ptr(arr: Optional[np.ndarray], ctype):
    if arr is None or arr.size == 0:
        return C.cast(None, C.POINTER(ctype))
    return arr.ctypes.data_as(C.POINTER(ctype))


@dataclass
class PackedScene:
    """Flat tables of one scene.  All arrays are C-contiguous and owned by this object."""

    obj_kind: np.ndarray          # (n_obj,) int32
    obj_index: np.ndarray         # (n_obj,) int32
    materials: np.ndarray         # (n_obj, 7) float64
    spheres: np.ndarray           # (n_sph, 4) float64
    sphere_obj: np.ndarray        # (n_sph,) int32
    mesh_aabb: np.ndarray         # (n_mesh, 6) float64
    mesh_obj: np.ndarray          # (n_mesh,) int32
    mesh_tri_begin: np.ndarray    # (n_mesh+1,) int32
    tri_verts: np.ndarray         # (n_tri, 9) float64
    cam_pos: tuple
    cam_right: tuple
    cam_up: tuple
    cam_forward: tuple
    cam_d: float
    cam_dof_strength: float
    cam_dof_dist: float
    sky_rgb: Optional[np.ndarray] = None   # (H, W, 3) uint8
    sky_scale: float = 1.0
    bvh_nodes: Optional[np.ndarray] = None      # structured (n,) rt_bvh_node
    bvh_tri_index: Optional[np.ndarray] = None  # (n_tri,) int32
    library_bvh: bool = False                   # no host hierarchy: ask the library to build one whatever the size
    object_names: list = field(default_factory=list)

    @property
    def n_objects(self) -> int:
        return int(self.obj_kind.shape[0])

    @property
    def n_spheres(self) -> int:
        return int(self.spheres.shape[0])

    @property
    def n_meshes(self) -> int:
        return int(self.mesh_aabb.shape[0])

    @property
    def n_tris(self) -> int:
        return int(self.tri_verts.shape[0])

    def desc(self) -> RtSceneDesc:
        """ctypes view of the tables; valid while ``self`` is alive."""
        d = RtSceneDesc()
        d.struct_size = C.sizeof(RtSceneDesc)
        d.n_objects = self.n_objects
        d.obj_kind = _ptr(self.obj_kind, C.c_int32)
        d.obj_index = _ptr(self.obj_index, C.c_int32)
        d.materials = _ptr(self.materials, C.c_double)
        d.n_spheres = self.n_spheres
        d.spheres = _ptr(self.spheres, C.c_double)
        d.sphere_obj = _ptr(self.sphere_obj, C.c_int32)
        d.n_meshes = self.n_meshes
        d.mesh_aabb = _ptr(self.mesh_aabb, C.c_double)
        d.mesh_obj = _ptr(self.mesh_obj, C.c_int32)
        d.mesh_tri_begin = _ptr(self.mesh_tri_begin, C.c_int32)
        d.n_tris = self.n_tris
        d.tri_verts = _ptr(self.tri_verts, C.c_double)
        d.cam_pos = (C.c_double * 3)(*self.cam_pos)
        d.cam_right = (C.c_double * 3)(*self.cam_right)
        d.cam_up = (C.c_double * 3)(*self.cam_up)
        d.cam_forward = (C.c_double * 3)(*self.cam_forward)
        d.cam_d = float(self.cam_d)
        d.cam_dof_strength = float(self.cam_dof_strength)
        d.cam_dof_dist = float(self.cam_dof_dist)
        if self.sky_rgb is not None and self.sky_rgb.size:
            d.sky_h, d.sky_w = int(self.sky_rgb.shape[0]), int(self.sky_rgb.shape[1])
            d.sky_rgb = _ptr(self.sky_rgb, C.c_uint8)
        else:
            d.sky_w = d.sky_h = 0
            d.sky_rgb = C.cast(None, C.POINTER(C.c_uint8))
        d.sky_scale = float(self.sky_scale)
        if self.bvh_nodes is not None and len(self.bvh_nodes):
            d.n_bvh_nodes = int(len(self.bvh_nodes))
            d.bvh_nodes = self.bvh_nodes.ctypes.data_as(C.c_void_p)
            d.bvh_tri_index = _ptr(self.bvh_tri_index, C.c_int32)
        else:
            d.n_bvh_nodes = -1 if self.library_bvh else 0
            d.bvh_nodes = None
            d.bvh_tri_index = C.cast(None, C.POINTER(C.c_int32))
        sw = layout_switches()
        d.flags, d.linear_max_tris = sw["flags"], sw["linear_max_tris"]
        d.bvh_leaf_size, d.build_threads = sw["bvh_leaf_size"], sw["build_threads"]
        return d

    def with_subdivided_mesh(self, mesh: int, levels: int) -> "PackedScene":
        """Copy of the scene in which every triangle of mesh ``mesh`` is split ``levels`` times into
        four by its edge midpoints (same surface, 4**levels times the faces; face order: for each
        parent in order its corner triangles at a, b, c, then the middle one).  Used to synthesise
        large meshes from small model files; any hierarchy is dropped."""
        import dataclasses

        b0, b1 = int(self.mesh_tri_begin[mesh]), int(self.mesh_tri_begin[mesh + 1])
        tv = self.tri_verts[b0:b1].reshape(-1, 3, 3)
        for _ in range(int(levels)):
            a, b, c = tv[:, 0], tv[:, 1], tv[:, 2]
            ab, bc, ca = (a + b) * 0.5, (b + c) * 0.5, (c + a) * 0.5
            tv = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                           np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
        new = np.ascontiguousarray(np.concatenate([self.tri_verts[:b0], tv.reshape(-1, 9), self.tri_verts[b1:]], 0))
        begin = self.mesh_tri_begin.copy()
        begin[mesh + 1:] += np.int32(tv.shape[0] - (b1 - b0))
        return dataclasses.replace(self, tri_verts=new, mesh_tri_begin=begin, bvh_nodes=None, bvh_tri_index=None)

    # -- persistence (fixtures / hand-off to a machine without the scene sources) ---------------
    def save(self, path: str, with_sky: bool = True) -> None:
        np.savez_compressed(
            path,
            obj_kind=self.obj_kind, obj_index=self.obj_index, materials=self.materials,
            spheres=self.spheres, sphere_obj=self.sphere_obj, mesh_aabb=self.mesh_aabb,
            mesh_obj=self.mesh_obj, mesh_tri_begin=self.mesh_tri_begin, tri_verts=self.tri_verts,
            camera=np.array([*self.cam_pos, *self.cam_right, *self.cam_up, *self.cam_forward,
                             self.cam_d, self.cam_dof_strength, self.cam_dof_dist], dtype=np.float64),
            sky_rgb=(self.sky_rgb if (with_sky and self.sky_rgb is not None) else np.zeros((0, 0, 3), np.uint8)),
            sky_scale=np.float64(self.sky_scale),
            object_names=np.array(self.object_names, dtype="U64"),
        )

    @classmethod
    def load(cls, path: str, sky_rgb: Optional[np.ndarray] = None) -> "PackedScene":
        z = np.load(path, allow_pickle=False)
        cam = z["camera"]
        sky = z["sky_rgb"]
        if sky.size == 0:
            sky = sky_rgb
        return cls(
            obj_kind=np.ascontiguousarray(z["obj_kind"], np.int32),
            obj_index=np.ascontiguousarray(z["obj_index"], np.int32),
            materials=np.ascontiguousarray(z["materials"], np.float64),
            spheres=np.ascontiguousarray(z["spheres"], np.float64).reshape(-1, 4),
            sphere_obj=np.ascontiguousarray(z["sphere_obj"], np.int32),
            mesh_aabb=np.ascontiguousarray(z["mesh_aabb"], np.float64).reshape(-1, 6),
            mesh_obj=np.ascontiguousarray(z["mesh_obj"], np.int32),
            mesh_tri_begin=np.ascontiguousarray(z["mesh_tri_begin"], np.int32),
            tri_verts=np.ascontiguousarray(z["tri_verts"], np.float64).reshape(-1, 9),
            cam_pos=tuple(cam[0:3]), cam_right=tuple(cam[3:6]), cam_up=tuple(cam[6:9]),
            cam_forward=tuple(cam[9:12]), cam_d=float(cam[12]), cam_dof_strength=float(cam[13]),
            cam_dof_dist=float(cam[14]),
            sky_rgb=(np.ascontiguousarray(sky, np.uint8) if sky is not None and sky.size else None),
            sky_scale=float(z["sky_scale"]),
            object_names=[str(s) for s in z["object_names"]],
        )


# ---------------------------------------------------------------------------------------------
# environment discovery
# ---------------------------------------------------------------------------------------------
def _surface_to_rgb(surface: Any) -> np.ndarray:
    """(H, W, 3) uint8 from anything offering get_width/get_height/get_at (a pygame Surface)."""
    if hasattr(surface, "rgb"):
        return np.ascontiguousarray(surface.rgb, dtype=np.uint8)
    w, h = int(surface.get_width()), int(surface.get_height())
    try:  # real pygame: one bulk copy
        import pygame  # type: ignore

        arr = pygame.surfarray.array3d(surface)  # (W, H, 3)
        return np.ascontiguousarray(np.transpose(arr, (1, 0, 2)), dtype=np.uint8)
    except Exception:
        out = np.empty((h, w, 3), dtype=np.uint8)
        for y in range(h):
            for x in range(w):
                c = surface.get_at((x, y))
                out[y, x] = (c[0], c[1], c[2])
        return out


def _probe_sky_scale(fn, rgb: np.ndarray) -> float:
    """Recover the constant factor a sky closure applies on top of texel/255 (0.4 in
    scenes/room.py:22, 1.0 in scenes/random_spheres.py:19-20) by evaluating it on a few rays."""
    from ..core import Ray, Vec3  # the closure only reads ray.direction

    h, w = rgb.shape[:2]
    dirs = [(1.0, 0.0, 0.0), (0.0, 0.6, 0.8), (-0.6, 0.0, 0.8), (0.48, -0.6, 0.64), (0.0, -0.8, -0.6)]
    for d in dirs:
        u = 0.5 + np.arctan2(d[2], d[0]) / (2 * np.pi)
        v = 1 - (0.5 + np.arcsin(d[1]) / np.pi)
        x, y = int(u * w), int(v * h)
        if not (0 <= x < w and 0 <= y < h):
            continue
        texel = rgb[y, x].astype(np.float64) / 255.0
        k = int(np.argmax(texel))
        if texel[k] <= 0:
            continue
        try:
            got = fn(Ray(Vec3(0), Vec3(*d)))
        except Exception:
            continue
        return float(tuple(got)[k] / texel[k])
    return 1.0


def _find_environment(scene: Any) -> tuple[Optional[np.ndarray], float]:
    sky = getattr(scene, "skybox", None)
    if sky is not None:  # this package's descriptor
        return np.ascontiguousarray(sky.rgb, dtype=np.uint8), float(sky.scale)
    fn = getattr(scene, "get_environment", None)
    glb = getattr(fn, "__globals__", None)
    if glb is not None and getattr(fn, "__self__", None) is None:
        surface = glb.get("sky_map")
        if surface is not None and hasattr(surface, "get_at"):
            rgb = _surface_to_rgb(surface)
            return rgb, _probe_sky_scale(fn, rgb)
    return None, 1.0


# ---------------------------------------------------------------------------------------------
def pack_scene(scene: Any, build_bvh: bool | None = None, sky_rgb: Optional[np.ndarray] = None,
               sky_scale: Optional[float] = None) -> PackedScene:
    """Flatten ``scene`` (this package's or upstream's) into a :class:`PackedScene`.

    ``build_bvh``: None = leave the choice to the library (linear table walk up to 4096 primitives,
    a hierarchy it builds itself beyond); True = build the hierarchy here with the native builder and
    hand it over; False = none from here.  ``sky_rgb`` (H, W, 3 uint8) / ``sky_scale`` override the
    environment found on the scene (needed when the scene's sky closure is opaque).
    """
    kinds, index, mats, names = [], [], [], []
    spheres, sphere_obj = [], []
    aabbs, mesh_obj, tri_begin, tri_chunks = [], [], [0], []
    n_tris = 0
    for oi, obj in enumerate(scene.objects):
        m = obj.material
        mats.append((*_vec(m.color), float(m.emission_strength), float(m.smoothness),
                     float(m.transmittance), float(m.ior)))
        names.append(str(getattr(obj, "name", "Object")))
        if hasattr(obj, "radius"):
            kinds.append(OBJ_SPHERE)
            index.append(len(spheres))
            spheres.append((*_vec(obj.origin), float(obj.radius)))
            sphere_obj.append(oi)
        elif hasattr(obj, "triangles"):
            kinds.append(OBJ_MESH)
            index.append(len(aabbs))
            tv = obj.transformed_vertices
            tv = np.asarray(tv if isinstance(tv, np.ndarray) else [tuple(v) for v in tv], dtype=np.float64).reshape(-1, 3)
            faces = np.asarray(obj.triangles, dtype=np.int64).reshape(-1, 3)
            bb = obj.bounding_box
            if bb is None:  # empty mesh: upstream's gate rejects every ray (mesh.py:66-67)
                aabbs.append((np.inf, np.inf, np.inf, -np.inf, -np.inf, -np.inf))
            else:
                aabbs.append((*_vec(bb[0]), *_vec(bb[1])))
            mesh_obj.append(oi)
            if faces.shape[0]:
                tri_chunks.append(tv[faces].reshape(-1, 9))
                n_tris += faces.shape[0]
            tri_begin.append(n_tris)
        else:
            raise TypeError(f"scene.objects[{oi}] is neither a sphere (radius) nor a mesh (triangles)")
    cam = scene.camera
    found_rgb, found_scale = _find_environment(scene)
    if sky_rgb is None:
        sky_rgb = found_rgb
    else:
        sky_rgb = np.ascontiguousarray(sky_rgb, dtype=np.uint8)
    if sky_scale is None:
        sky_scale = found_scale
    packed = PackedScene(
        obj_kind=np.asarray(kinds, dtype=np.int32),
        obj_index=np.asarray(index, dtype=np.int32),
        materials=np.asarray(mats, dtype=np.float64).reshape(-1, 7),
        spheres=np.asarray(spheres, dtype=np.float64).reshape(-1, 4),
        sphere_obj=np.asarray(sphere_obj, dtype=np.int32),
        mesh_aabb=np.asarray(aabbs, dtype=np.float64).reshape(-1, 6),
        mesh_obj=np.asarray(mesh_obj, dtype=np.int32),
        mesh_tri_begin=np.asarray(tri_begin, dtype=np.int32),
        tri_verts=(np.ascontiguousarray(np.concatenate(tri_chunks, axis=0)) if tri_chunks
                   else np.zeros((0, 9), dtype=np.float64)),
        cam_pos=_vec(cam.pos), cam_right=_vec(cam.right), cam_up=_vec(cam.up), cam_forward=_vec(cam.forward),
        cam_d=float(cam.d), cam_dof_strength=float(cam.dof_strength), cam_dof_dist=float(cam.dof_dist),
        sky_rgb=sky_rgb, sky_scale=sky_scale, object_names=names,
    )
    if build_bvh and packed.n_tris:
        from .bvh import build_flat_bvh

        packed.bvh_nodes, packed.bvh_tri_index = build_flat_bvh(packed.tri_verts)
    return packed
