"""Print how rt_set_scene lays the golden scenes out (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as h
from raytracing_amd.backend.lib import HipBackend

for name in sys.argv[1:] or ["room", "room_horse"]:
    b = HipBackend(0)
    b.set_scene(h.packed_scene(name))
    print(name, b.scene_layout(), flush=True)
