"""Dev tool (GPU box): frame and tile-kernel time of every k_raster_span shape (FRR_RASTER_NW / FRR_RASTER_OCC) on the
headline, 4096^2 and 4K-sheets workloads, with an image hash per shape (all shapes must agree).
  python tools/exp_shapes.py [headline|cfg4|cfg5 ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f_renderer_amd as fr
from f_renderer_amd import scenes

SHAPES = [(0, 0), (4, 6), (4, 8), (3, 6), (6, 6), (8, 6), (16, 4)]


def workload(name):
    if name == "headline":
        return 1920, 1080, scenes.random_clip_triangles(1_000_000, 1920, 1080), fr.VS_CLIP, fr.PS_DEPTH
    if name == "cfg4":
        return 4096, 4096, scenes.random_clip_triangles(1_000_000, 4096, 4096), fr.VS_CLIP, fr.PS_DEPTH
    if name == "cfg5":
        return 3840, 2160, scenes.layered_sheets(), fr.VS_PHONG, fr.PS_BLINN
    if name == "cfg3":
        return 1920, 1080, scenes.displaced_sphere(), fr.VS_PHONG, fr.PS_PHONG
    raise SystemExit(name)


for name in (sys.argv[1:] or ["headline", "cfg4", "cfg5"]):
    W, H, mesh, vs, ps = workload(name)
    for nw, occ in SHAPES:
        for k, v in (("FRR_RASTER_NW", nw), ("FRR_RASTER_OCC", occ)):
            if v:
                os.environ[k] = str(v)
            else:
                os.environ.pop(k, None)
        r = fr.Renderer(W, H)
        if ps != fr.PS_DEPTH:
            eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
            r.set_texture(0, scenes.checker_texture(1024, 32))
            r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
        m = r.upload_mesh(mesh, vs)
        r.set_count_fragments(False)
        for _ in range(3):
            r.clear(); r.draw(m, ps)
        r.sync()
        K = 20
        r.event_record(0)
        for _ in range(K):
            r.clear(); r.draw(m, ps)
        r.event_record(1)
        ms = r.event_elapsed_ms(0, 1) / K
        c, d, t = r.readback()
        h = hashlib.sha256(c.tobytes() + d.tobytes() + t.tobytes()).hexdigest()[:12]
        r.profile_enable(True, kernels=["k_raster"]); r.profile_reset()
        for _ in range(10):
            r.clear(); r.draw(m, ps)
        kt, kn = r.profile_get("k_raster")
        print(f"{name:9s} nw={nw:2d} occ={occ}: frame {ms*1e3:8.1f} us  k_raster {kt/max(kn,1)*1e3:8.1f} us  hash {h}", flush=True)
        r.close()
