"""Pool size (PRT_POOL_CAP) in adaptive mode (10..50 spp) on C4: device time per frame."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
p = api.default_params(10, 1234, max_spp=50)
for cap in (None, 256, 384, 512, 768, 1024):
    for blocks in (None, 5):
        if cap is None: os.environ.pop("PRT_POOL_CAP", None)
        else: os.environ["PRT_POOL_CAP"] = str(cap)
        if blocks is None: os.environ.pop("PRT_POOL_BLOCKS_PER_CU", None)
        else: os.environ["PRT_POOL_BLOCKS_PER_CU"] = str(blocks)
        r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
        cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(2)]
        ms = min(c.render_ms for c in cs)
        print("cap %s blocks/CU %s: %.2f ms, %d rays, %.0f Mrays/s" % (cap, blocks, ms, cs[0].ray_count, cs[0].ray_count / ms / 1e3), flush=True)
