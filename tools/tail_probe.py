"""How ragged is the end of a frame?  Mean of the waves' main-loop time over the longest wave's (k_pool, counting build,
PRT_DEBUG_UTIL): fixed 8 spp and adaptive 10..50 spp on C4, full frame and the 1/8 shard."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PRT_DEBUG_UTIL"] = "1"
import torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for spp, mx in ((8, 0), (10, 50)):
    for nr in (1, 8):
        p = api.default_params(spp, 1234, max_spp=mx, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS)
        sys.stderr.write("spp %d max %d shard 1/%d\n" % (spp, mx, nr)); sys.stderr.flush()
        c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
