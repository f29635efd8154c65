"""Adaptive mode (10..50 spp) on C4: the 4-waves-per-SIMD kernel (128 VGPRs, default) against a 5-wave build of the same
kernel (96 VGPRs, PRT_ADAPT_WAVES5=1).  Same process, same box; contexts are created after the variable is set."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
imgs = {}
for rep in range(2):
    for v in ("", "1"):
        if v: os.environ["PRT_ADAPT_WAVES5"] = v
        else: os.environ.pop("PRT_ADAPT_WAVES5", None)
        r = api.Renderer(0); r.upload(hs)
        p = api.default_params(10, 1234, max_spp=50, pipeline=capi.PIPELINE_POOL)
        r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
        cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(3)]
        ms = min(c.render_ms for c in cs)
        imgs[v] = buf.cpu().numpy().copy()
        print("waves/SIMD %s: %.2f ms  %d rays  %.0f Mrays/s" % ("5" if v else "4", ms, cs[0].ray_count, cs[0].ray_count / ms / 1e3), flush=True)
        del r
print("images bit-identical:", bool((imgs[""].view(np.uint32) == imgs["1"].view(np.uint32)).all()))
