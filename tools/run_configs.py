#!/usr/bin/env python3
"""All five BASELINE.json configs (+ the headline) at FULL size on one MI355X: GPU frame time
(clear + draw, HIP events, statistic off), the 1-core CPU oracle's time for the same frame, and a
full-size bit-exact comparison of depth / triangle ids / RGBA8 against the oracle.

  python tools/run_configs.py [--json profiles/r01_configs.json] [--skip-oracle]

The oracle is used here exactly as in the tests: as the checker and the timed CPU baseline."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import f_renderer_amd as fr  # noqa: E402
from f_renderer_amd import scenes  # noqa: E402


def camera(W, H):
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    return eye, fr.set_look_at(eye, at, up), fr.set_perspective(fovy, aspect, zn, zf)


def configs():
    tex = scenes.checker_texture(1024, 32)
    yield dict(name="cfg1 single triangle 512x512 flat", W=512, H=512, mesh=scenes.single_triangle(), vs="CLIP", ps="FLAT", cam=False, tex=None)
    yield dict(name="cfg2 torus 6,272 tris 1920x1080 Gouraud", W=1920, H=1080, mesh=scenes.torus(), vs="GOURAUD", ps="COLOR", cam=True, tex=None)
    yield dict(name="cfg3 sphere 69,192 tris 1920x1080 Phong textured", W=1920, H=1080, mesh=scenes.displaced_sphere(), vs="PHONG", ps="PHONG", cam=True, tex=tex)
    yield dict(name="cfg3b sphere 69,192 tris 1920x1080 Blinn-Phong textured", W=1920, H=1080, mesh=scenes.displaced_sphere(), vs="PHONG", ps="BLINN", cam=True, tex=tex)
    yield dict(name="cfg4 1M random tris 4096x4096 depth-only", W=4096, H=4096, mesh=scenes.random_clip_triangles(1_000_000, 4096, 4096), vs="CLIP", ps="DEPTH", cam=False, tex=None)
    yield dict(name="cfg5 sheets 250,000 tris 3840x2160 Blinn-Phong textured", W=3840, H=2160, mesh=scenes.layered_sheets(), vs="PHONG", ps="BLINN", cam=True, tex=tex)
    yield dict(name="headline 1M random tris 1920x1080 depth-only", W=1920, H=1080, mesh=scenes.random_clip_triangles(1_000_000, 1920, 1080), vs="CLIP", ps="DEPTH", cam=False, tex=None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--only", default=None, help="run only the configs whose name contains this string")
    args = ap.parse_args()
    rows = []
    for cfg in configs():
        if args.only and args.only not in cfg["name"]:
            continue
        W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
        vs, ps = getattr(fr, "VS_" + cfg["vs"]), getattr(fr, "PS_" + cfg["ps"])
        r = fr.Renderer(W, H)
        kw = {}
        if cfg["cam"]:
            eye, view, proj = camera(W, H)
            kw = dict(view=view, proj=proj, view_pos=eye)
        if cfg["tex"] is not None:
            r.set_texture(0, cfg["tex"])
            kw["texture_slot"] = 0
        r.set_uniforms(flat_color=(1.0, 0.5, 0.25, 1.0), **kw)
        m = r.upload_mesh(mesh, vs)
        ntris = mesh.shape[0]
        r.set_count_fragments(True)
        r.clear(); r.draw(m, ps); r.sync()
        st = r.stats()
        c_g, d_g, t_g = r.readback()
        r.set_count_fragments(False)
        for _ in range(3):
            r.clear(); r.draw(m, ps)
        r.sync()
        r.event_record(0)
        for _ in range(args.frames):
            r.clear(); r.draw(m, ps)
        r.event_record(1)
        ms = r.event_elapsed_ms(0, 1) / args.frames
        c2, d2, t2 = r.readback()
        same_fast = bool(np.array_equal(c2, c_g) and np.array_equal(d2.view(np.uint32), d_g.view(np.uint32)) and np.array_equal(t2, t_g))
        row = dict(config=cfg["name"], width=W, height=H, triangles=ntris, setup_triangles=st["tris_setup"],
                   covered_fragments=st["frag_covered"], gpu_ms=round(ms, 4), gpu_mtri_s=round(ntris / ms / 1e3, 2),
                   gpu_mfrag_s=round(st["frag_covered"] / ms / 1e3, 1), earlyz_output_identical=same_fast)
        r.profile_enable(True); r.profile_reset()
        for _ in range(5):
            r.clear(); r.draw(m, ps)
        kus = {}
        for k in fr.Renderer.KERNELS:
            t, cnt = r.profile_get(k)
            if cnt:
                kus[k] = round(t / cnt * 1e3, 1)
        r.profile_enable(False)
        row.update(bin_entries=st["bin_entries"], kernels_us=kus)
        if not args.skip_oracle:
            from oracle import cref
            okw = {}
            if cfg["cam"]:
                eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
                okw = dict(view=cref.set_look_at(eye, at, up), proj=cref.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
            if cfg["tex"] is not None:
                okw["tex"] = cref.Texture(cfg["tex"])
            u = cref.make_uniforms(flat_color=(1.0, 0.5, 0.25, 1.0), **okw)
            f = cref.Frame(W, H)
            t0 = time.perf_counter()
            f.clear()
            f.draw(mesh, getattr(cref, "VS_" + cfg["vs"]), getattr(cref, "PS_" + cfg["ps"]), u)
            cpu_s = time.perf_counter() - t0
            oc = f.counters.as_dict()
            row.update(cpu_1core_s=round(cpu_s, 3), cpu_mtri_s=round(ntris / cpu_s / 1e6, 4),
                       speedup=round((ntris / ms / 1e3) / (ntris / cpu_s / 1e6), 1),
                       parity_depth_bits=bool(np.array_equal(d_g.view(np.uint32), f.depth.view(np.uint32))),
                       parity_triangle_ids=bool(np.array_equal(t_g, f.tri_id)),
                       parity_rgba8=bool(np.array_equal(c_g, f.color)) if ps != fr.PS_DEPTH else None,
                       parity_counts=bool(oc["frag_covered"] == st["frag_covered"] and oc["tris_setup"] == st["tris_setup"]),
                       frag_zpass=oc["frag_zpass"], frag_nan=oc["frag_nan"])
        print(json.dumps(row), flush=True)
        rows.append(row)
        r.close()
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
