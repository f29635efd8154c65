"""Frame time of C4 under PRT_RESERVE_CUS=8 with different CU-mask patterns (PRT_RESERVE_PATTERN), one process per setting."""
import sys, os, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import sys, os, tempfile
sys.path.insert(0, %r)
import torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
ms = sorted(r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True).render_ms for _ in range(12))[:-2]
print("%%.3f / %%.3f ms (min / median of 10)" %% (ms[0], ms[len(ms) // 2]))
''' % root
for rounds in range(2):
    for res, pat in ((0, 0), (8, 0), (8, 1), (8, 2), (8, 3), (16, 2), (32, 2)):
        env = dict(os.environ)
        env.pop("PRT_RESERVE_CUS", None); env.pop("PRT_RESERVE_PATTERN", None)
        if res: env["PRT_RESERVE_CUS"] = str(res); env["PRT_RESERVE_PATTERN"] = str(pat)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print("reserve %2d pattern %d: %s" % (res, pat, out[-1] if out else "failed"), flush=True)
