"""Wavefront pipeline on a C4 frame under PRT_CHAINS / PRT_SHADE_BLOCK / PRT_TRACE_BLOCKS_PER_CU settings."""
import sys, os, tempfile, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
p = api.default_params(8, 1234, pipeline=2)
def run(reps=5):
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    return min(r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True).render_ms for _ in range(reps))
for chains, sb, bpc in itertools.product((1, 2, 3), (256, 1024), (None, 3, 4, 5)):
    os.environ["PRT_CHAINS"] = str(chains); os.environ["PRT_SHADE_BLOCK"] = str(sb)
    if bpc is None: os.environ.pop("PRT_TRACE_BLOCKS_PER_CU", None)
    else: os.environ["PRT_TRACE_BLOCKS_PER_CU"] = str(bpc)
    print("chains %d shade_block %4d trace blocks/CU %s: %.2f ms" % (chains, sb, bpc, run()), flush=True)
