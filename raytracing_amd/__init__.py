"""MI355X-native path-tracing backend with the scene / camera / material / renderer API of the
m-J-K-b/raytracing project.

Sub-modules mirror upstream's ``src`` package layout (src/__init__.py:1-4):

    from raytracing_amd import core, mesh, raytracing, post_processing

``core`` and ``mesh`` are pure scene description; ``raytracing.Renderer`` drives hand-written HIP
kernels (gfx950) through the C ABI declared in include/raytracing_amd.h.  There is no CPU render
path in this package: rendering without the compiled library raises.
"""
from __future__ import annotations

import importlib

__version__ = "0.1.0"
__all__ = ["core", "mesh", "raytracing", "post_processing", "scenes", "backend"]


def __getattr__(name):
    if name in __all__:
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
