"""Wavefront vs pool pipeline by scene size: terrains of growing triangle count, 1920x1080x8spp full frame and 1/4 frame."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
w, h = 1920, 1080
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for quads, tiles in ((128, 8), (256, 16), (384, 16), (512, 32), (708, 32)):
    s = scenes.terrain(quads, tiles, size=float(quads)); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
    hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
    r = api.Renderer(0); r.upload(hs)
    cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
    def run(pipeline, nranks, reps=4):
        p = api.default_params(8, 1234, pipeline=pipeline)
        r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr())
        return min(r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr()).render_ms for _ in range(reps))
    print("terrain %d^2 (%d tris): full wavefront %.2f pool %.2f | 1/2 %.2f %.2f | 1/4 %.2f %.2f" % (
        quads, 2 * quads * quads, run(2, 1), run(4, 1), run(2, 2), run(4, 2), run(2, 4), run(4, 4)), flush=True)
    r.close()
