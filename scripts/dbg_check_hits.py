"""Runs one GPU parity test with _check_hits replaced by a version that prints the worst ray (developer tool):
    python scripts/dbg_check_hits.py test_box_rows 1"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import pytest
import test_gpu_parity as T
from raytracing_amd.backend.lib import HipBackend

last = {}
orig_intersect = HipBackend.intersect
def intersect(self, o, d, origin_prim=None):
    last.update(o=np.asarray(o), d=np.asarray(d), g=origin_prim)
    return orig_intersect(self, o, d, origin_prim=origin_prim)
HipBackend.intersect = intersect

def check(got, want, tol=T.TOL):
    hit = want["obj"] >= 0
    rel = np.where(hit, np.abs(got["t"] - want["t"]) / np.where(hit, want["t"], 1.0), 0.0)
    idx = np.argsort(-rel)[:5]
    np.set_printoptions(precision=17)
    for i in idx:
        if rel[i] > 2e-6:
            print(f"ray {i}: rel {rel[i]:.3e} got t {got['t'][i]!r} want {want['t'][i]!r} obj {want['obj'][i]} prim {want['prim'][i]} gprim {want['gprim'][i]} how {got['how'][i]}"
                  f"\n   o {last['o'][i]!r}\n   d {last['d'][i]!r} origin {None if last['g'] is None else last['g'][i]}", flush=True)
T._check_hits = check
name = sys.argv[1]
args = [int(a) for a in sys.argv[2:]]
be = HipBackend(0)
mp = pytest.MonkeyPatch()
fn = getattr(T, name)
import inspect
params = inspect.signature(fn).parameters
kw = {}
if "backend" in params: kw["backend"] = be
if "monkeypatch" in params: kw["monkeypatch"] = mp
rest = [p for p in params if p not in kw]
for p, a in zip(rest, args): kw[p] = a
fn(**kw)
