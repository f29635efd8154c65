"""max |dRGB| and ray counts of every golden fixture on the GPU (default pipeline choice + both main pipelines)."""
import sys, os, glob
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import camera_and_params, load_golden, scene_dir
from par_raytracer_amd import api
cache = {}
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    name = os.path.basename(path)[:-4]
    if name == "kat": continue
    g = load_golden(name)
    key = (str(g["scene"]), int(g["light_mode"]))
    if key not in cache:
        s, d = scene_dir(key[0])
        r = api.Renderer(0); r.upload(api.HostScene(d, "scene.obj", key[1], s.camera_position)); cache[key] = r
    r = cache[key]
    line = "%-28s" % name
    adaptive = "max_spp" in g and int(g["max_spp"]) > int(g["spp"])
    for pl in ((4,) if adaptive else (2, 4)):
        cam, p = camera_and_params(g, pl)
        img, c = r.render_lattice(cam, p, int(g["width"]), int(g["height"]), int(g["lattice"]))
        d = np.abs(img[:, :, :3] - g["rgb"])
        line += "  %s: rays %s max|d| %.3g" % ({2: "wavefront", 4: "pool"}[pl], "equal" if c.ray_count == int(g["ray_count"]) else "DIFFER", d.max())
    print(line, flush=True)
