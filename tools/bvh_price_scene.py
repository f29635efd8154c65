"""Dumps a generated scene's triangles + camera for tools/bvh_price.cpp: python3 tools/bvh_price_scene.py terrain_1m out.bin"""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from par_raytracer_amd import scenes

s = scenes.make_scene(sys.argv[1])
tris = []
for g in s.groups:
    tris.append(s.positions[g.faces[:, :, 0]].reshape(-1, 9))
v = np.concatenate(tris).astype(np.float32)
with open(sys.argv[2], "wb") as f:
    f.write(struct.pack("<I", len(v)))
    f.write(v.tobytes())
    f.write(struct.pack("<7f", *s.camera_position, *s.camera_facing, s.fov))
print(len(v), "triangles")
