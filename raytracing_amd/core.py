"""Host-side value types of the renderer API: ``Vec3``, ``Ray``, ``HitInfo``, ``Material``.

These mirror the names and fields of the upstream data model so that scene-building code written
against it keeps working (upstream: src/core/vec3.py:7-26, src/core/ray.py:6-13,
src/core/hit_info.py:11-24, src/core/material.py:8-34, src/core/util.py:6-32).  Upstream derives
``Vec3`` from ``pygame.Vector3``; here it is a self-contained three-double value type that offers
the subset of that interface the scene code uses, because the GPU backend only ever reads the
components when it packs a scene.  No per-ray arithmetic happens in these classes.
"""
from __future__ import annotations

import copy as _copy
import math
from typing import Iterable, Tuple

import numpy as np

__all__ = ["Vec3", "Ray", "HitInfo", "Material", "util", "lerp", "smoothstep",
           "smooth_interpolation", "exponential_interpolation", "quadratic_formula"]


def _is_scalar(v) -> bool:
    return isinstance(v, (int, float, np.integer, np.floating))


class _ComponentwiseView:
    """Result of ``Vec3.elementwise()``: ``view * other`` multiplies component by component."""

    __slots__ = ("_v",)

    def __init__(self, v: "Vec3") -> None:
        self._v = v

    def __mul__(self, other):
        v = self._v
        if _is_scalar(other):
            return Vec3(v.x * other, v.y * other, v.z * other)
        ox, oy, oz = other
        return Vec3(v.x * ox, v.y * oy, v.z * oz)

    __rmul__ = __mul__


class Vec3:
    """Three doubles with the vector interface the scene code relies on.

    Accepted constructor forms: ``Vec3()``, ``Vec3(s)`` (broadcast), ``Vec3(x, y, z)``,
    ``Vec3(sequence_of_three)``.
    """

    __slots__ = ("x", "y", "z")

    def __init__(self, *args) -> None:
        n = len(args)
        if n == 3:
            self.x, self.y, self.z = float(args[0]), float(args[1]), float(args[2])
        elif n == 1:
            a = args[0]
            if _is_scalar(a):
                self.x = self.y = self.z = float(a)
            else:
                ax, ay, az = a
                self.x, self.y, self.z = float(ax), float(ay), float(az)
        elif n == 0:
            self.x = self.y = self.z = 0.0
        else:
            raise TypeError(f"Vec3 takes 0, 1 or 3 positional arguments, got {n}")

    # -- container protocol (numpy and tuple() go through it) --------------------------------
    def __iter__(self):
        yield self.x
        yield self.y
        yield self.z

    def __len__(self) -> int:
        return 3

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __array__(self, dtype=None, copy=None):
        return np.array((self.x, self.y, self.z), dtype=dtype or np.float64)

    def __repr__(self) -> str:
        return f"Vec3({self.x!r}, {self.y!r}, {self.z!r})"

    def __eq__(self, other) -> bool:
        try:
            ox, oy, oz = other
        except Exception:
            return False
        return self.x == ox and self.y == oy and self.z == oz

    __hash__ = None  # mutable value type

    # -- arithmetic ---------------------------------------------------------------------------
    def __add__(self, o):
        ox, oy, oz = o
        return Vec3(self.x + ox, self.y + oy, self.z + oz)

    __radd__ = __add__

    def __iadd__(self, o):
        ox, oy, oz = o
        self.x += ox
        self.y += oy
        self.z += oz
        return self

    def __sub__(self, o):
        ox, oy, oz = o
        return Vec3(self.x - ox, self.y - oy, self.z - oz)

    def __rsub__(self, o):
        ox, oy, oz = o
        return Vec3(ox - self.x, oy - self.y, oz - self.z)

    def __neg__(self):
        return Vec3(-self.x, -self.y, -self.z)

    def __mul__(self, o):
        if _is_scalar(o):
            return Vec3(self.x * o, self.y * o, self.z * o)
        return self.dot(o)  # vector * vector is the scalar product, as in pygame

    __rmul__ = __mul__

    def __truediv__(self, s):
        return Vec3(self.x / s, self.y / s, self.z / s)

    def __itruediv__(self, s):
        self.x /= s
        self.y /= s
        self.z /= s
        return self

    # -- vector operations --------------------------------------------------------------------
    def dot(self, o) -> float:
        ox, oy, oz = o
        return self.x * ox + self.y * oy + self.z * oz

    def cross(self, o) -> "Vec3":
        ox, oy, oz = o
        return Vec3(self.y * oz - self.z * oy, self.z * ox - self.x * oz, self.x * oy - self.y * ox)

    def length_squared(self) -> float:
        return self.x * self.x + self.y * self.y + self.z * self.z

    def magnitude(self) -> float:
        return math.sqrt(self.length_squared())

    length = magnitude

    def normalize(self) -> "Vec3":
        l = self.magnitude()
        if l == 0.0:
            raise ValueError("Can't normalize Vector of length Zero")
        return Vec3(self.x / l, self.y / l, self.z / l)

    def reflect(self, normal) -> "Vec3":
        n = Vec3(normal)
        l2 = n.length_squared()
        if l2 == 0.0:
            raise ValueError("Normal must not be of length zero.")
        if l2 != 1.0:
            n = n / math.sqrt(l2)
        k = 2.0 * self.dot(n)
        return Vec3(self.x - k * n.x, self.y - k * n.y, self.z - k * n.z)

    def rotate(self, degrees: float, axis) -> "Vec3":
        """Rotate by ``degrees`` about ``axis`` (Rodrigues' formula)."""
        if degrees == 0:
            return self.copy()
        k = Vec3(axis).normalize()
        a = math.radians(degrees)
        c, s = math.cos(a), math.sin(a)
        kxv = k.cross(self)
        kd = k.dot(self) * (1.0 - c)
        return Vec3(self.x * c + kxv.x * s + k.x * kd,
                    self.y * c + kxv.y * s + k.y * kd,
                    self.z * c + kxv.z * s + k.z * kd)

    def copy(self) -> "Vec3":
        return Vec3(self.x, self.y, self.z)

    def elementwise(self) -> _ComponentwiseView:
        return _ComponentwiseView(self)

    # -- the three helpers upstream adds on top of pygame.Vector3 (vec3.py:11-26) ---------------
    @classmethod
    def random_unit(cls) -> "Vec3":
        """Normalised point of the cube [-1,1)^3 drawn from numpy's global generator.

        This is NOT uniform on the sphere (no rejection step); scene-building code depends on the
        exact draw order (three ``np.random.random()`` calls: x, y, z), so it is kept as is.
        """
        rx = np.random.random() * 2 - 1
        ry = np.random.random() * 2 - 1
        rz = np.random.random() * 2 - 1
        return cls(rx, ry, rz).normalize()

    def absolute(self) -> "Vec3":
        """Make the components non-negative IN PLACE and return a copy (upstream quirk)."""
        self.x, self.y, self.z = abs(self.x), abs(self.y), abs(self.z)
        return self.copy()

    def prod(self, other) -> "Vec3":
        ox, oy, oz = other
        return Vec3(self.x * ox, self.y * oy, self.z * oz)


class Ray:
    """Origin, direction (not necessarily unit) and the per-path inside/outside toggle."""

    __slots__ = ("origin", "direction", "inside")

    def __init__(self, origin: Vec3, direction: Vec3, inside: bool = False) -> None:
        self.origin = origin
        self.direction = direction
        self.inside = inside

    def at(self, d: float) -> Vec3:
        return self.origin + self.direction * d


class HitInfo:
    """One ray/object intersection record; ``obj`` is the scene object that was hit."""

    __slots__ = ("obj", "hit", "depth", "pos", "normal", "primitive")

    def __init__(self, obj: object, hit: bool = False, depth: float = 0.0,
                 pos: Vec3 | None = None, normal: Vec3 | None = None, primitive: int = -1) -> None:
        self.obj = obj
        self.hit = hit
        self.depth = depth
        self.pos = pos
        self.normal = normal
        self.primitive = primitive  # extension: face index / sphere root; upstream has no such field


class Material:
    """colour, emission_strength, smoothness, transmittance, ior — seven floats on the device."""

    def __init__(self, color: Vec3, emission_strength: float, smoothness: float,
                 transmittance: float, ior: float) -> None:
        self.color = color
        self.emission_strength = emission_strength
        self.smoothness = smoothness
        self.transmittance = transmittance
        self.ior = ior

    @classmethod
    def default_material(cls) -> "Material":
        return cls(color=Vec3(1), emission_strength=0.0, smoothness=0.0, transmittance=0.0, ior=1.45)

    def copy(self) -> "Material":
        return _copy.copy(self)

    def as_row(self) -> Tuple[float, ...]:
        c = self.color
        return (float(c[0]), float(c[1]), float(c[2]), float(self.emission_strength),
                float(self.smoothness), float(self.transmittance), float(self.ior))


# -- scalar helpers (src/core/util.py:6-32) ------------------------------------------------------
def smoothstep(v, minv, maxv):
    if v < minv:
        return 0
    if v > maxv:
        return 1
    s = (v - minv) / (maxv - minv)
    return s * s * (3 - 2 * s)


def lerp(v1, v2, t):
    return v1 + (v2 - v1) * t


def smooth_interpolation(v1, v2, t):
    return lerp(v1, v2, smoothstep(t, 0, 1))


def exponential_interpolation(v1, v2, t, exponent: float = 0.5):
    return v1 + (v2 - v1) * min(max(t, 0), 1) ** exponent


def quadratic_formula(a: float, b: float, c: float) -> Tuple[float, float]:
    root = (b * b - 4 * a * c) ** 0.5
    return (-b - root) / (2 * a), (-b + root) / (2 * a)


class _UtilNamespace:
    """``core.util`` as upstream exposes it (src/core/__init__.py:1)."""

    smoothstep = staticmethod(smoothstep)
    lerp = staticmethod(lerp)
    smooth_interpolation = staticmethod(smooth_interpolation)
    exponential_interpolation = staticmethod(exponential_interpolation)
    quadratic_formula = staticmethod(quadratic_formula)


util = _UtilNamespace()
