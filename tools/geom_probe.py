"""Dev tool (GPU box): geometry-kernel time of the Phong vertex shader on meshes of different character (all visible /
config 5's sheets / everything off screen / everything clipped), to see what the kernel's time depends on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes

W, H = 3840, 2160
eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
sheets = scenes.layered_sheets()
n = sheets.shape[0]
rng = np.random.default_rng(1)
def small(center_scale, size):
    c = rng.uniform(-center_scale, center_scale, (n, 1, 3)).astype(np.float32); c[..., 2] *= 0.2
    v = c + rng.uniform(-size, size, (n, 3, 3)).astype(np.float32)
    m = np.zeros((n, 3, 8), np.float32); m[..., :3] = v; m[..., 3:5] = rng.uniform(0, 1, (n, 3, 2)); m[..., 5:] = (0, 0, 1)
    return m
cases = {"sheets (cfg5)": sheets, "all visible, small": small(0.8, 0.01), "all off screen": small(0.5, 0.01) + np.array([50, 0, 0, 0, 0, 0, 0, 0], np.float32),
         "sheets shuffled": sheets[rng.permutation(n)], "sheets inner layer x5": np.concatenate([sheets[: n // 5]] * 5)}
for name, mesh in cases.items():
    r = fr.Renderer(W, H)
    r.set_texture(0, scenes.checker_texture(64, 8))
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    m = r.upload_mesh(np.ascontiguousarray(mesh, np.float32), fr.VS_PHONG)
    for _ in range(3):
        r.clear(); r.draw(m, fr.PS_BLINN)
    st = r.stats()
    r.profile_enable(True, kernels=["k_geom", "k_bin_seg", "k_raster"]); r.profile_reset()
    for _ in range(10):
        r.clear(); r.draw(m, fr.PS_BLINN)
    out = []
    for k in ("k_geom", "k_bin_seg", "k_raster"):
        t, c = r.profile_get(k); out.append(f"{k[2:]} {t / max(c, 1) * 1e3:6.1f}")
    print(f"{name:24s} tris {n}  setup {st['tris_setup']:7d} | " + " | ".join(out), flush=True)
    r.close()
