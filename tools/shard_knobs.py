"""1/8 shard (and the full frame) of C4 under pool knobs, shared and private pools, one process (set_option), min of n renders.
    python tools/shard_knobs.py"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
def run(nr, n=8):
    r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    ms = sorted(r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True).render_ms for _ in range(n))
    return ms[0], ms[len(ms) // 2]
for shared in (0, 1):
    r.set_option("POOL_SHARED", shared)
    for cap in (None, 64, 128, 192, 256, 320, 448, 512):
        for topup in (None, 64, 128):
            r.set_option("POOL_CAP", cap)
            r.set_option("POOL_TOPUP", topup)
            a = run(8); b = run(1, 4)
            print("shared %d cap %-5s topup %-5s  1/8 shard min %.3f med %.3f   full min %.3f med %.3f" % (shared, cap, topup, a[0], a[1], b[0], b[1]), flush=True)
r.set_option("POOL_CAP", None); r.set_option("POOL_TOPUP", None)
for shared in (0, 1):
    r.set_option("POOL_SHARED", shared)
    for bpc in (None, 4, 3):
        r.set_option("POOL_BLOCKS_PER_CU", bpc)
        a = run(8); b = run(1, 4)
        print("shared %d blocks/CU %-5s  1/8 shard min %.3f med %.3f   full min %.3f med %.3f" % (shared, bpc, a[0], a[1], b[0], b[1]), flush=True)
