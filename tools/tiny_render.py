"""One small render through the C ABI with everything on stderr visible: tools/tiny_render.py <scene> <pipeline> [w h spp]"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from par_raytracer_amd import api, capi, scenes
name, pipeline = sys.argv[1], int(sys.argv[2])
w, h, spp = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (32, 24, 2)
s = scenes.make_scene(name); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
r = api.Renderer(0)
info = r.upload(hs)
print("uploaded: %d tris, %d nodes (%d B each), depth %d" % (info.triangle_count, info.bvh_node_count, info.bvh_node_bytes, info.bvh_max_depth), file=sys.stderr, flush=True)
img, c = r.render(cam, api.default_params(spp, 1234, pipeline=pipeline | capi.FLAG_COUNT_VISITS), w, h)
print("rendered: rays %d nodes %d tris %d, %.3f ms, mean rgb %s" % (c.ray_count, c.node_visits, c.tri_tests, c.render_ms, img[:, :3].mean(axis=0)), file=sys.stderr, flush=True)
r.close()
