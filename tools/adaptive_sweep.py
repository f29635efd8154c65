import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
def run(p):
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    return min(r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True).render_ms for _ in range(2))
print("fixed 50 spp, pool: %.1f ms" % run(api.default_params(50, 1234, pipeline=4)), flush=True)
os.environ["PRT_DEBUG_UTIL"] = "1"
for cap in (256, 512, 1024, 2048):
    os.environ["PRT_POOL_CAP"] = str(cap)
    p = api.default_params(10, 1234, max_spp=50)
    print("adaptive cap %d: %.1f ms" % (cap, run(p)), flush=True)
    pc = api.default_params(10, 1234, max_spp=50, pipeline=capi.FLAG_COUNT_VISITS)
    r.render_device(cam, pc, w, h, 0, w * h, buf.data_ptr(), True)
