"""Tests-only stand-in for the subset of ``pygame`` that the upstream path tracer touches.

This module exists for ONE purpose: to let ``tools/gen_golden.py`` import the unmodified
upstream Python renderer in the build container (where the real ``pygame`` wheel is not
installed) so that golden vectors can be generated from the upstream code itself.
It is never imported by the product package, never shipped to a GPU box, and it is not a
re-implementation of pygame: only ``Vector3`` (plain IEEE-double component arithmetic),
``Surface`` (a size holder) and ``image.load`` (PIL-backed, nearest-texel ``get_at``) exist.

Arithmetic note: every operation is the textbook component formula evaluated in Python
floats (IEEE double, no fused multiply-add), which is what pygame's C ``Vector3`` does as
well; last-ulp differences against the real C code (e.g. divide vs multiply-by-reciprocal
in ``normalize``) are possible and are far below every tolerance used by the tests.
"""
from __future__ import annotations

import math

from . import image  # noqa: F401  (pygame exposes pygame.image)


class _Elementwise:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v

    def __mul__(self, o):
        v = self.v
        if isinstance(o, (int, float)):
            return type(v)(v.x * o, v.y * o, v.z * o)
        return type(v)(v.x * o[0], v.y * o[1], v.z * o[2])

    __rmul__ = __mul__


class Vector3:
    __slots__ = ("x", "y", "z")

    def __init__(self, *a):
        if len(a) == 0:
            self.x = self.y = self.z = 0.0
        elif len(a) == 1:
            s = a[0]
            if isinstance(s, (int, float)):
                self.x = self.y = self.z = float(s)
            else:
                self.x, self.y, self.z = float(s[0]), float(s[1]), float(s[2])
        elif len(a) == 3:
            self.x, self.y, self.z = float(a[0]), float(a[1]), float(a[2])
        else:
            raise TypeError("Vector3 takes 0, 1 or 3 arguments")

    # sequence protocol (numpy converts instances through it)
    def __len__(self):
        return 3

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __repr__(self):
        return f"{type(self).__name__}({self.x!r}, {self.y!r}, {self.z!r})"

    def _new(self, x, y, z):
        return type(self)(x, y, z)

    def __add__(self, o):
        return self._new(self.x + o[0], self.y + o[1], self.z + o[2])

    __radd__ = __add__

    def __iadd__(self, o):
        self.x += o[0]
        self.y += o[1]
        self.z += o[2]
        return self

    def __sub__(self, o):
        return self._new(self.x - o[0], self.y - o[1], self.z - o[2])

    def __rsub__(self, o):
        return self._new(o[0] - self.x, o[1] - self.y, o[2] - self.z)

    def __neg__(self):
        return self._new(-self.x, -self.y, -self.z)

    def __mul__(self, o):
        if isinstance(o, (int, float)) or not hasattr(o, "__len__"):
            o = float(o)
            return self._new(self.x * o, self.y * o, self.z * o)
        return self.x * o[0] + self.y * o[1] + self.z * o[2]  # vec*vec is the dot product

    __rmul__ = __mul__

    def __truediv__(self, s):
        s = float(s)
        return self._new(self.x / s, self.y / s, self.z / s)

    def __itruediv__(self, s):
        s = float(s)
        self.x /= s
        self.y /= s
        self.z /= s
        return self

    def __eq__(self, o):
        try:
            return self.x == o[0] and self.y == o[1] and self.z == o[2]
        except Exception:
            return False

    __hash__ = object.__hash__

    def dot(self, o):
        return self.x * o[0] + self.y * o[1] + self.z * o[2]

    def cross(self, o):
        return self._new(
            self.y * o[2] - self.z * o[1],
            self.z * o[0] - self.x * o[2],
            self.x * o[1] - self.y * o[0],
        )

    def length_squared(self):
        return self.x * self.x + self.y * self.y + self.z * self.z

    def magnitude(self):
        return math.sqrt(self.length_squared())

    length = magnitude

    def normalize(self):
        l = math.sqrt(self.length_squared())
        if l == 0:
            raise ValueError("Can't normalize Vector of length Zero")
        return self._new(self.x / l, self.y / l, self.z / l)

    def reflect(self, n):
        nl2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2]
        if nl2 == 0:
            raise ValueError("Normal must not be of length zero.")
        nx, ny, nz = n[0], n[1], n[2]
        if nl2 != 1:
            nl = math.sqrt(nl2)
            nx, ny, nz = nx / nl, ny / nl, nz / nl
        d = self.x * nx + self.y * ny + self.z * nz
        return self._new(self.x - 2 * nx * d, self.y - 2 * ny * d, self.z - 2 * nz * d)

    def rotate(self, deg, axis):
        if deg == 0:
            return self._new(self.x, self.y, self.z)
        al = math.sqrt(axis[0] ** 2 + axis[1] ** 2 + axis[2] ** 2)
        kx, ky, kz = axis[0] / al, axis[1] / al, axis[2] / al
        a = math.radians(deg)
        c, s = math.cos(a), math.sin(a)
        kd = kx * self.x + ky * self.y + kz * self.z
        cx = ky * self.z - kz * self.y
        cy = kz * self.x - kx * self.z
        cz = kx * self.y - ky * self.x
        return self._new(
            self.x * c + cx * s + kx * kd * (1 - c),
            self.y * c + cy * s + ky * kd * (1 - c),
            self.z * c + cz * s + kz * kd * (1 - c),
        )

    def copy(self):
        return self._new(self.x, self.y, self.z)

    def elementwise(self):
        return _Elementwise(self)


class Surface:
    """Size holder; the oracle harness never blits or saves."""

    def __init__(self, size=(0, 0), *a, **k):
        self._size = (int(size[0]), int(size[1]))

    def get_width(self):
        return self._size[0]

    def get_height(self):
        return self._size[1]


class math_ns:  # pygame.math.Vector3 alias
    Vector3 = Vector3


import sys as _sys

_sys.modules[__name__ + ".math"] = math_ns  # type: ignore
