"""Sweep the pool pipeline's knobs (env vars read per render call) on C4: full frame and the 1/8 shard."""
import sys, time, os, tempfile, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

def run(pipeline, nranks, reps=4):
    p = api.default_params(8, 1234, pipeline=pipeline)
    r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr())
    ms = []
    for _ in range(reps):
        c = r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr())
        ms.append(c.render_ms)
    return min(ms), float(np.mean(ms))

def sweep(label, envs):
    keys = list(envs)
    for vals in itertools.product(*[envs[k] for k in keys]):
        for k, v in zip(keys, vals):
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = str(v)
        a = run(4, 1); b = run(4, 8)
        print("%s %s: full %.2f (mean %.2f)  1/8 shard %.2f (mean %.2f)" % (label, dict(zip(keys, vals)), a[0], a[1], b[0], b[1]), flush=True)
    for k in keys: os.environ.pop(k, None)

print("wavefront: full %.2f  1/8 shard %.2f" % (run(2, 1)[0], run(2, 8)[0]), flush=True)
which = sys.argv[1:] or ["cap", "keep", "blocks"]
if "cap" in which: sweep("cap", {"PRT_POOL_CAP": [64, 128, 256, 512, 1024, 2048]})
if "keep" in which: sweep("keep", {"PRT_KEEP_MIN": [24, 32, 40, 48], "PRT_NODE_MIN": [16, 32]})
if "keep2" in which: sweep("keep2", {"PRT_KEEP_MIN": [36, 40, 44, 48, 56], "PRT_NODE_MIN": [24, 32, 40, 48]})
if "frac" in which: sweep("frac", {"PRT_NODE_FRAC": [4, 2, 3, 5, 6, 7, 4], "PRT_NODE_MIN": [32, 64]})
if "shard" in which: sweep("shard", {"PRT_POOL_BLOCKS_PER_CU": [5, 4, 3], "PRT_POOL_CAP": [None, 128, 192, 256, 320, 384]})
if "blocks" in which: sweep("blocks", {"PRT_POOL_BLOCKS_PER_CU": [1, 2]})
if "topup" in which: sweep("topup", {"PRT_POOL_CAP": [256, 512], "PRT_POOL_TOPUP": [64, 128, 256]})
if "default" in which: sweep("default", {"PRT_POOL_NOP": [0]})
if "stagger" in which: sweep("stagger", {"PRT_POOL_STAGGER": [0, 1], "PRT_POOL_CAP": [256, 512], "PRT_POOL_TOPUP": [64, None]})
