"""Render driver API: ``Camera``, ``Scene``, ``RenderSettings``, ``RenderResult``, ``Renderer``, ``util``.

Same class names, constructor signatures and polled attributes as upstream's ``src.raytracing``
package (camera.py:8-50, scene.py:13-46, render_settings.py:6-22, render_result.py:7-55,
renderer.py:14-150), so an application written against it (upstream's main.py) runs unchanged.
What differs is where the work happens: ``Renderer.render`` packs the scene once and drives the
HIP kernels through the C ABI; nothing here traces rays on the CPU.

Additions (keyword arguments with defaults, never positional):
  ``Renderer(device=0, seed=0, samples_per_launch=None)``, ``Scene.skybox`` / ``Scene.set_skybox``,
  ``Scene.intersect_batch``, ``RenderResult.tensor()`` (zero-copy torch view of the device buffer).
"""
from __future__ import annotations

import math
import threading
from threading import Event, Thread
from typing import List, Optional, Tuple

import numpy as np

from .core import HitInfo, Ray, Vec3

__all__ = ["Camera", "Scene", "RenderSettings", "RenderResult", "Renderer", "Skybox", "util"]


# ---------------------------------------------------------------------------------------------
# shading helpers (src/raytracing/util.py:11-35) — kept for scene code / tests that call them
# ---------------------------------------------------------------------------------------------
class util:  # noqa: N801  (upstream exposes a module of this name)
    @staticmethod
    def schlick_approximation(incident: Vec3, normal: Vec3, ior: float) -> float:
        cos = incident.dot(normal)
        r0 = ((1 - ior) / (1 + ior)) ** 2
        return r0 + (1 - r0) * (1 - cos) ** 5

    @staticmethod
    def refract(incident: Vec3, normal: Vec3, etai_over_etat: float) -> Vec3:
        cos_theta = min(normal.dot(-incident), 1)
        perp = (incident + normal * cos_theta) * etai_over_etat
        par = normal * (-math.sqrt(abs(1.0 - perp.length_squared())))
        return perp + par

    @staticmethod
    def vec_to_sky_coords(vec) -> Tuple[float, float]:
        x, y, z = vec
        return 0.5 + math.atan2(z, x) / (2 * math.pi), 1 - (0.5 + math.asin(y) / math.pi)

    @staticmethod
    def random_hemisphere_sample(normal: Vec3) -> Vec3:
        nd = Vec3.random_unit()
        return -nd if nd.dot(normal) < 0 else nd


# ---------------------------------------------------------------------------------------------
class Camera:
    """Pinhole camera with optional square-aperture depth of field (camera.py:9-37)."""

    def __init__(self, pos: Vec3, fov: float, look_at: Vec3, dof_strength: float = 0, dof_dist: float = 1) -> None:
        self.pos = pos
        self.fov = fov
        self.d = 1 / np.tan(fov / 2)
        self.look_at = look_at
        self.update_axis()
        self.dof_strength = dof_strength
        self.dof_dist = dof_dist

    def set_fov(self, fov: float) -> None:
        self.fov = fov
        self.d = 1 / np.tan(fov / 2)

    def set_look_at(self, look_at: Vec3) -> None:
        self.look_at = look_at
        self.update_axis()

    def update_axis(self) -> None:
        self.forward = (Vec3(self.look_at) - Vec3(self.pos)).normalize()
        self.right = Vec3(0, 1, 0).cross(self.forward).normalize()
        self.up = self.forward.cross(self.right).normalize()

    def get_ray(self, u: float, v: float) -> Ray:
        """The pinhole ray through screen coordinates (u, v); depth-of-field jitter is applied on
        the device per sample and is not reproduced here."""
        direction = (self.right * u + self.up * v + self.forward * float(self.d)).normalize()
        return Ray(Vec3(self.pos), direction)


class Skybox:
    """Equirectangular environment: nearest texel / 255 * scale (scenes/room.py:15-22)."""

    def __init__(self, rgb: np.ndarray, scale: float = 1.0) -> None:
        rgb = np.asarray(rgb)
        if rgb.ndim != 3 or rgb.shape[2] < 3:
            raise ValueError("skybox image must be (H, W, 3) uint8")
        self.rgb = np.ascontiguousarray(rgb[:, :, :3], dtype=np.uint8)
        self.scale = float(scale)

    @classmethod
    def from_file(cls, path: str, scale: float = 1.0) -> "Skybox":
        from PIL import Image

        with Image.open(path) as im:
            return cls(np.asarray(im.convert("RGB")), scale)

    def sample(self, direction) -> Vec3:
        u, v = util.vec_to_sky_coords(direction)
        h, w = self.rgb.shape[:2]
        x, y = min(max(int(u * w), 0), w - 1), min(max(int(v * h), 0), h - 1)
        return Vec3(*(self.rgb[y, x].astype(np.float64) / 255.0 * self.scale))


class Scene:
    def __init__(self) -> None:
        self.objects: list = []
        self.camera: Camera = Camera(Vec3(0), np.pi / 3, Vec3(0, 0, 1))
        self.skybox: Optional[Skybox] = None
        self._backend = None
        self._packed_key = None

    def add_object(self, obj) -> None:
        self.objects.append(obj)

    # -- environment ----------------------------------------------------------------------------
    def set_skybox(self, rgb_or_skybox, scale: float = 1.0) -> None:
        self.skybox = rgb_or_skybox if isinstance(rgb_or_skybox, Skybox) else Skybox(rgb_or_skybox, scale)

    def set_environment(self, img) -> None:
        """Upstream's intended semantics (scene.py:29-32): image / max(image), indexed by (u, v)."""
        arr = np.asarray(img)
        if arr.ndim == 3 and arr.shape[2] >= 3:
            arr = arr[:, :, :3].astype(np.float64)
            peak = float(arr.max()) or 1.0
            self.skybox = Skybox(np.clip(np.round(arr / peak * 255.0), 0, 255).astype(np.uint8), 1.0)
        else:
            raise ValueError("environment image must be (H, W, 3)")

    def get_environment(self, ray: Ray) -> Vec3:
        if self.skybox is None:
            return Vec3(0)
        return self.skybox.sample(ray.direction)

    # -- closest hit (GPU) ----------------------------------------------------------------------
    def fingerprint(self) -> tuple:
        """What the device copy of this scene depends on, cheap to compute (no vertex is read):
        object identities and order, material values, sphere centres and radii, each mesh's change
        counter (``Mesh.version``: every setter - mesh.py:178-209 - and every assignment to its
        arrays bumps it), camera and environment.  A changed fingerprint means the tables on the GPU
        are stale.  Array identities are NOT used: the setters free one array and allocate the next,
        and CPython hands the same address out again.  Mesh-like objects without a change counter
        (upstream's own ``Mesh``) are keyed on their transform values, vertex and face counts and
        the bounding box their setters recompute."""
        def mat(m):
            return (tuple(m.color), float(m.emission_strength), float(m.smoothness), float(m.transmittance), float(m.ior))

        def vec(v):
            return None if v is None else tuple(float(x) for x in v)

        objs = []
        for o in self.objects:
            if hasattr(o, "radius"):
                objs.append((id(o), mat(o.material), tuple(o.origin), float(o.radius)))
            elif hasattr(o, "version"):
                objs.append((id(o), mat(o.material), int(o.version)))
            else:
                bb = getattr(o, "bounding_box", None)
                objs.append((id(o), mat(o.material), vec(getattr(o, "origin", None)), vec(getattr(o, "rotation", None)),
                             vec(getattr(o, "scale", None)), len(o.transformed_vertices), len(o.triangles),
                             None if bb is None else (vec(bb[0]), vec(bb[1]))))
        c = self.camera
        cam = (tuple(c.pos), tuple(c.right), tuple(c.up), tuple(c.forward), float(c.d), float(c.dof_strength), float(c.dof_dist))
        sky = None if self.skybox is None else (id(self.skybox.rgb), float(self.skybox.scale))
        return (tuple(objs), cam, sky)

    def update(self, t) -> None:
        """Advance every object's animations to time ``t`` (base.py:17-19); the next query or render
        re-packs the scene if that moved anything."""
        for o in self.objects:
            o.update(t)

    def _device_scene(self, device: int = 0):
        from .backend.lib import HipBackend
        from .backend.packer import pack_scene

        key = (self.fingerprint(), device)
        if self._backend is None or self._packed_key != key:
            be = self._backend if (self._backend is not None and self._backend.device == device) else HipBackend(device)
            # (refit: a scene that only moved keeps its hierarchy's topology; the library checks that it is the same meshes)
            be.set_scene(pack_scene(self), refit=True)
            self._backend, self._packed_key = be, key
        return self._backend

    def invalidate(self) -> None:
        """Force a re-pack on the next query (for edits the fingerprint cannot see, e.g. vertex
        arrays modified in place)."""
        self._packed_key = None

    def intersect_batch(self, origins, directions, origin_prim=None) -> dict:
        """Closest hit of many rays on the GPU (see ``HipBackend.intersect``)."""
        return self._device_scene().intersect(origins, directions, origin_prim)

    def intersect(self, ray: Ray) -> List[HitInfo]:
        """``Scene.intersect(ray)`` as the renderer consumes it: a list holding the closest hit (or
        nothing).  Upstream returns every hit sorted by depth but only ever reads element 0
        (renderer.py:36)."""
        r = self.intersect_batch([tuple(ray.origin)], [tuple(ray.direction)])
        if r["obj"][0] < 0:
            return []
        obj = self.objects[int(r["obj"][0])]
        return [HitInfo(obj=obj, hit=True, depth=float(r["t"][0]), pos=Vec3(r["pos"][0]), normal=Vec3(r["normal"][0]),
                        primitive=int(r["prim"][0]))]


class RenderSettings:
    def __init__(self, scene: Scene, width: int, height: int, render_passes: int, max_bounces: int) -> None:
        self.WIDTH: int = width
        self.HEIGHT: int = height
        self.ASPECT: float = height / width
        self.RENDER_PASSES: int = render_passes
        self.MAX_BOUNCES: int = max_bounces
        self.SCENE: Scene = scene
        self.AREA: List[float] = [0, 1, 0, 1]
        self.EPS = 0.0


class RenderResult:
    """Accumulated radiance and progress counters, as the application polls them
    (render_result.py:14-55).  ``arr`` is the un-normalised sum over finished passes, (W, H, 3),
    x-major, y up; the display views are y-flipped and divided by ``rendered_passes``."""

    def __init__(self, width: int, height: int, render_passes: int) -> None:
        self.RENDER_PASSES = render_passes
        self.WIDTH, self.HEIGHT = width, height
        self.PX_NUM = width * height
        self.arr = np.zeros((width, height, 3), dtype=np.float64)
        self.rendered_passes = 0
        self.rendered_pixels = 0
        self.finished = False
        self.img_arr = np.zeros_like(self.arr)
        self.img_arr_clipped = np.zeros_like(self.arr)
        self.surface = None
        self.stats: dict = {}
        self.error: Optional[BaseException] = None   # what ended a background render abnormally
        self.failed = False                          # terminal like `finished`, but without an image
        self._backend = None
        self._stale = False                          # render_area launched work `arr` does not show yet
        self._lock = threading.Lock()

    def add_to_px(self, x: int, y: int, v) -> None:
        if 0 <= x < self.arr.shape[0] and 0 <= y < self.arr.shape[1]:
            self.arr[x, y] += np.asarray(tuple(v))
            self.rendered_pixels += 1

    @property
    def progress(self) -> float:
        return self.rendered_pixels / self.PX_NUM / self.RENDER_PASSES

    @property
    def progress_percent(self) -> float:
        return self.progress * 100

    def update_views(self) -> None:
        if self._stale and self._backend is not None:   # tiles rendered through render_area since the last read-back
            self._backend.synchronize()
            arr = self._backend.read().astype(np.float64)
            with self._lock:
                self.arr = arr
                self._stale = False
        self.update_img_arr()
        self.update_img_arr_clipped()

    def update_img_arr(self) -> None:
        with self._lock:
            n = max(self.rendered_passes, 1)
            np.copyto(self.img_arr, self.arr[:, ::-1] / n)

    def update_img_arr_clipped(self) -> None:
        with self._lock:
            n = max(self.rendered_passes, 1)
            np.copyto(self.img_arr_clipped, np.clip(self.arr[:, ::-1] / n, 0, 1))

    def update_surface(self) -> None:
        try:
            import pygame as pg  # optional: only the interactive preview needs it

            self.surface = pg.surfarray.make_surface(self.img_arr_clipped * 255)
        except ImportError:
            self.surface = None

    def save(self, path) -> None:
        """Write the clipped view as an image file (PIL; pygame is not required)."""
        from PIL import Image

        self.update_views()
        img = np.transpose(np.clip(self.img_arr_clipped * 255.0, 0, 255).astype(np.uint8), (1, 0, 2))
        Image.fromarray(img, "RGB").save(path)

    def tensor(self):
        """Zero-copy ``torch`` view (W, H, 3) float32 of the device accumulation buffer, for
        device-side post-processing.  Valid until the renderer is re-initialised."""
        import torch

        be = self._backend
        if be is None or not be.accum_ptr():
            raise RuntimeError("no device buffer: render first")

        class _Iface:
            pass

        holder = _Iface()
        holder.__cuda_array_interface__ = {"shape": (self.WIDTH, self.HEIGHT, 3), "typestr": "<f4",
                                           "data": (be.accum_ptr(), False), "version": 2}
        return torch.as_tensor(holder, device=f"cuda:{be.device}")


class Renderer:
    """Drop-in for upstream's ``Renderer`` (renderer.py:14-150) backed by the HIP kernels."""

    def __init__(self, device: int = 0, seed: int = 0, samples_per_launch: Optional[int] = None) -> None:
        self.scene: Scene = None  # type: ignore
        self.render_result: RenderResult = None  # type: ignore
        self.render_settings: RenderSettings = None  # type: ignore
        self.stop_event = Event()
        self.device = device
        self.seed = seed
        self.samples_per_launch = samples_per_launch
        self._backend = None
        self._thread: Optional[Thread] = None
        self._raytrace_calls = 0
        self._scene_key = None
        self._tile_passes: dict = {}   # render_area: passes already launched per tile

    # -- set-up ---------------------------------------------------------------------------------
    def init_render(self, render_settings: RenderSettings) -> None:
        from .backend.lib import HipBackend
        from .backend.packer import pack_scene

        if self._thread is not None and self._thread.is_alive():   # a render of this renderer is still running
            self.stop_render()
            self._thread.join()
        self.render_settings = render_settings
        self.render_result = RenderResult(render_settings.WIDTH, render_settings.HEIGHT, render_settings.RENDER_PASSES)
        self.scene = render_settings.SCENE
        self.stop_event.clear()
        self._tile_passes = {}
        # One context per renderer, kept across renders; the scene tables are uploaded again only
        # when the scene changed (objects moved by the mesh setters, materials, camera, environment).
        be = self._backend if self._backend is not None else HipBackend(self.device)
        key = (id(self.scene), self.scene.fingerprint())
        if self._scene_key != key:
            be.set_scene(pack_scene(self.scene), refit=True)
            self._scene_key = key
        be.clear_stop()
        be.reset_stats()
        be.resize(render_settings.WIDTH, render_settings.HEIGHT)
        self._backend = be
        self.render_result._backend = be

    def divide_area(self, vertical_splits: int, horizontal_splits: int) -> List[Tuple[int, int, int, int]]:
        """(x0, x1, y0, y1) tiles; the last tile of each axis absorbs the remainder."""
        W, H = self.render_settings.WIDTH, self.render_settings.HEIGHT
        w, h = W // vertical_splits, H // horizontal_splits
        return [(i * w, W if i == vertical_splits - 1 else (i + 1) * w,
                 j * h, H if j == horizontal_splits - 1 else (j + 1) * h)
                for i in range(vertical_splits) for j in range(horizontal_splits)]

    # -- rendering ------------------------------------------------------------------------------
    def _launch(self, area, sample_begin: int, sample_count: int) -> None:
        s = self.render_settings
        self._backend.render(s.WIDTH, s.HEIGHT, sample_begin, sample_count, s.MAX_BOUNCES, self.seed, rect=area)

    def _publish(self, passes_done: int, pixels_done: int) -> None:
        res = self.render_result
        arr = self._backend.read().astype(np.float64)
        with res._lock:
            res.arr = arr
            res.rendered_passes = passes_done
            res.rendered_pixels = pixels_done

    def render_area(self, area: Tuple[int, int, int, int]) -> None:
        """One pass (one more sample per pixel) over ``area``; upstream's per-tile entry point
        (renderer.py:90-106).  Each tile keeps its own count of passes, which is the sample index of
        its next call: a caller that drives the tiles itself (upstream's loop: all tiles, then
        ``rendered_passes += 1``; or any other order) gets a fresh sample per call and tile, never
        the same one twice.  The host copy ``render_result.arr`` is NOT refreshed here - a
        whole-frame read-back per tile would dominate - but by ``Renderer.refresh()`` or
        ``RenderResult.update_views()``."""
        if self.stop_event.is_set():
            return
        res = self.render_result
        area = tuple(int(v) for v in area)
        with res._lock:
            sample = self._tile_passes.get(area, res.rendered_passes)
            self._tile_passes[area] = sample + 1
        self._launch(area, sample, 1)
        x0, x1, y0, y1 = area
        with res._lock:
            res.rendered_pixels += (x1 - x0) * (y1 - y0)
            res._stale = True

    def refresh(self) -> None:
        """Bring ``render_result.arr`` up to date with the device buffer (waits for launched work)."""
        res = self.render_result
        self._backend.synchronize()
        arr = self._backend.read().astype(np.float64)
        with res._lock:
            res.arr = arr
            res._stale = False

    def render(self, vertical_splits: int = 1, horizontal_splits: int = 1) -> None:
        """All passes over the whole image.  The tile arguments are accepted for compatibility; the
        device schedules (pixel, sample-chunk) jobs itself, so tiles buy nothing here.

        Runs on the background thread ``start_render`` starts, so it must not die silently: a stop
        request that lands between the check and the launch comes back from the library as
        RT_ERR_STOPPED and ends the loop cleanly; any other library error is kept in
        ``render_result.error`` and ``render_result.failed`` is set, which pollers treat as terminal."""
        from .backend.lib import RT_ERR_STOPPED, BackendError

        s, res = self.render_settings, self.render_result
        total = s.RENDER_PASSES
        step = self.samples_per_launch or total
        done = 0
        try:
            while done < total and not self.stop_event.is_set():
                n = min(step, total - done)
                self._launch((0, s.WIDTH, 0, s.HEIGHT), done, n)
                self._backend.synchronize()
                done += n
                self._publish(done, done * res.PX_NUM)
            res.stats = self._backend.stats()
        except BackendError as e:
            if e.status != RT_ERR_STOPPED:
                res.error = e
                res.failed = True
        except BaseException as e:   # anything else on this thread would otherwise vanish with it
            res.error = e
            res.failed = True
            raise
        finally:
            res.finished = done >= total and not res.failed

    def start_executor(self, vertical_splits, horizontal_splits) -> None:
        self._thread = Thread(target=self.render, args=(vertical_splits, horizontal_splits), daemon=True)
        self._thread.start()

    def start_render(self, render_settings: RenderSettings, vertical_splits: int = 1, horizontal_splits: int = 1,
                     ) -> RenderResult:
        self.init_render(render_settings)
        if self.samples_per_launch is None:  # progressive by default when driven asynchronously
            self.samples_per_launch = max(1, render_settings.RENDER_PASSES // 16)
        self.start_executor(vertical_splits, horizontal_splits)
        return self.render_result

    def stop_render(self) -> None:
        self.stop_event.set()
        if self._backend is not None:
            self._backend.request_stop()

    def is_rendering(self) -> bool:
        """True while the background thread of ``start_render`` is alive."""
        return self._thread is not None and self._thread.is_alive()

    def wait(self, timeout: Optional[float] = None) -> None:
        if self._thread is not None:
            self._thread.join(timeout)

    # -- single rays / ray batches ---------------------------------------------------------------
    def _path_backend(self):
        if self.render_settings is None:
            raise RuntimeError("Renderer.raytrace needs render_settings (MAX_BOUNCES) and a scene: call init_render first "
                               "or set .scene and .render_settings")
        scene = self.scene if self.scene is not None else getattr(self.render_settings, "SCENE", None)
        if self._backend is not None and scene is None:
            return self._backend   # a context handed in ready-made: nothing to compare it with
        if self._backend is not None:
            # the renderer's own context: re-upload if the scene was edited since init_render
            from .backend.packer import pack_scene

            key = (id(scene), scene.fingerprint())
            if self._scene_key != key:
                self._backend.set_scene(pack_scene(scene), refit=True)
                self._scene_key = key
            return self._backend
        return scene._device_scene(self.device)

    def raytrace_batch(self, origins, directions, inside=None, depth: int = 0, sample: Optional[int] = None) -> np.ndarray:
        """Radiance (n, 3) float32 of the paths that start with the given rays, traced on the GPU:
        upstream's ``raytrace`` (renderer.py:21-79) for many rays at once.  Every call draws from a
        fresh random stream unless ``sample`` pins it."""
        if sample is None:
            sample, self._raytrace_calls = self._raytrace_calls, self._raytrace_calls + 1
        return self._path_backend().raytrace(origins, directions, self.render_settings.MAX_BOUNCES, self.seed, inside=inside,
                                             first_depth=depth, sample=sample)

    def raytrace(self, ray: Ray, incoming_light=None, ray_color=None, depth=None):
        """``incoming_light + ray_color * L(ray)`` as upstream returns it (renderer.py:21-79); the
        path is traced on the GPU, one ray per call (use ``raytrace_batch`` for many)."""
        incoming_light = Vec3(0, 0, 0) if incoming_light is None else incoming_light
        ray_color = Vec3(1, 1, 1) if ray_color is None else ray_color
        depth = 0 if depth is None else int(depth)
        if depth > self.render_settings.MAX_BOUNCES:
            return Vec3(0, 0, 0)
        L = self.raytrace_batch([tuple(ray.origin)], [tuple(ray.direction)], inside=[bool(getattr(ray, "inside", False))],
                                depth=depth)[0]
        return incoming_light + Vec3(*(float(ray_color[k]) * float(L[k]) for k in range(3)))
