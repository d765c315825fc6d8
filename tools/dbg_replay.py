import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes
from oracle import cref
W, H = 200, 150
tris = scenes.random_clip_triangles(6000, W, H, seed=8, spread=1.3, w_jitter=0.5)
small = scenes.random_clip_triangles(50, W, H, seed=9, spread=0.5)
f = cref.Frame(W, H); f.clear()
f.draw(small, cref.VS_CLIP, cref.PS_DEPTH, cref.make_uniforms())
f.draw(tris, cref.VS_CLIP, cref.PS_DEPTH, cref.make_uniforms())
for fan_cap, count in ((0, False), (64, False), (64, True)):
    r = fr.Renderer(W, H)
    r.set_option("fan_capacity", fan_cap)
    m0, m = r.upload_mesh(small, fr.VS_CLIP), r.upload_mesh(tris, fr.VS_CLIP)
    r.clear()
    r.draw(m0, fr.PS_DEPTH)
    print("after draw0", r.stats())
    n = r.geometry_processing(m, count=count)
    print("after geom", n, r.stats())
    r.rasterization((0, W), (0, H), fr.PS_DEPTH)
    print("after raster", r.stats())
    _, d, t = r.readback()
    dt = t.astype(np.int64) - f.tri_id.astype(np.int64)
    print("fan_cap", fan_cap, "count", count, "id diffs", np.unique(dt)[:8], "oracle setup", f.counters.tris_setup, flush=True)
    r.close()
