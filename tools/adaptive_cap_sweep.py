"""Adaptive mode (10..50 spp) on C4: pool capacity per wave (PRT_POOL_CAP) against frame time.  With the default capacity all
pixels of the frame are handed out at the start; a smaller pool leaves pixels on the counter for waves whose own pixels end
early (sky: 10 samples of one ray each)."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
ref = None
for cap, topup in ((0, 0), (128, 16), (128, 32), (128, 48), (128, 80), (128, 96), (128, 112), (128, 128), (0, 0), (128, 32), (128, 96)):
    for k, v in (("PRT_POOL_CAP", cap), ("PRT_POOL_TOPUP", topup)):
        if v: os.environ[k] = str(v)
        else: os.environ.pop(k, None)
    r = api.Renderer(0); r.upload(hs)
    p = api.default_params(10, 1234, max_spp=50, pipeline=capi.PIPELINE_POOL)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(2)]
    ms = min(c.render_ms for c in cs)
    img = buf.cpu().numpy().view(np.uint32).copy()
    if ref is None: ref = img
    print("cap %4s top-up at %4s free: %.2f ms  %d rays  %.0f Mrays/s  image identical: %s" % (cap or "auto", topup or "auto", ms, cs[0].ray_count, cs[0].ray_count / ms / 1e3, bool((img == ref).all())), flush=True)
    del r
