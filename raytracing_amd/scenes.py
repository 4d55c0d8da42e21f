"""The two benchmark scenes, built with this package's classes (upstream: scenes/room.py:25-52,
scenes/random_spheres.py:23-96).  Asset files are NOT shipped here; pass the directory that holds
upstream's ``assets/`` tree (default ``./assets``, as upstream's relative paths assume)."""
from __future__ import annotations

import os

import numpy as np

from .core import Material, Vec3
from .mesh import Mesh, Sphere
from .raytracing import Camera, Scene, Skybox

__all__ = ["room", "random_spheres"]


def _sky(asset_root: str, scale: float):
    path = os.path.join(asset_root, "skyboxes", "skybox1.png")
    return Skybox.from_file(path, scale) if os.path.exists(path) else None


def room(asset_root: str = "./assets") -> Scene:
    """Open cube of five coloured walls with a box lamp and a glossy sphere; camera at z = -2."""
    objects = Mesh.load_from_obj_file(os.path.join(asset_root, "models", "obj", "room.obj"))
    palette = {"left": (Vec3(1, 0.5, 0.5), 0), "back": (Vec3(0.5, 1, 0.5), 0), "right": (Vec3(0.5, 0.5, 1), 0),
               "lamp": (Vec3(1), 1)}
    for o in objects:
        color, emission = palette.get(o.name, (Vec3(1), 0))
        o.material = Material(color, emission, 0, 0, 1)
    objects.append(Sphere(Material(Vec3(1), 0, 0.3, 0.1, 1.4), Vec3(0), 0.5))
    scene = Scene()
    scene.objects = objects
    scene.camera = Camera(Vec3(0, 0, -2), np.pi / 2, Vec3(0))
    scene.skybox = _sky(asset_root, 0.4)
    return scene


def random_spheres(asset_root: str = "./assets", count: int = 200) -> Scene:
    """Ground sphere (r = 1e5), light sphere (r = 5000) and ``count`` non-overlapping spheres on a
    disc of radius 100: a third metal, a third emissive, a third glass.  Draws from numpy's global
    generator in upstream's order, so ``np.random.seed(s)`` reproduces upstream's scene for seed s."""
    scene = Scene()
    scene.skybox = _sky(asset_root, 1.0)

    def plain(emission):
        m = Material.default_material()
        m.color, m.smoothness, m.emission_strength, m.transmittance = Vec3(1), 0, emission, 0
        return m

    scene.add_object(Sphere(plain(0), Vec3(0, -100000, 0), 100000))
    scene.add_object(Sphere(plain(1), Vec3(5000), 5000))
    positions, radii = [], []
    while len(positions) < count:
        p = Vec3.random_unit().elementwise() * Vec3(1, 0, 1) * 100
        r = 1 / (np.random.random() + 0.95) ** 17 + 1.5
        if any((q - p).magnitude() < r + rq for q, rq in zip(positions, radii)):
            continue
        positions.append(p)
        radii.append(r)
    for p, r in zip(positions, radii):
        p.y += r
        m = Material.default_material()
        kind = np.random.random()
        if kind < 1 / 3:      # metal
            m.color = Vec3.random_unit().absolute()
            m.smoothness = np.random.random() * 0.8 + 0.1
            m.emission_strength, m.transmittance = 0, 0
        elif kind < 2 / 3:    # emissive
            m.color = Vec3.random_unit().absolute()
            m.smoothness = 0
            m.emission_strength = np.random.random() + 0.5
            m.transmittance = 0
        else:                 # glass
            m.color = Vec3(1)
            m.smoothness, m.emission_strength = 0, 0
            m.transmittance = np.random.random() * 0.5 + 0.5
        scene.add_object(Sphere(m, p, r))
    scene.camera = Camera(Vec3(0, 0, 5), np.pi / 2, Vec3(0, 0, 0), dof_strength=0.01, dof_dist=10)
    return scene
