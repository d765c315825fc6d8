#!/usr/bin/env python3
"""Writes tests/golden/frames.json: SHA-256 digests of what the C oracle (oracle/frr_oracle.c) produces for every
BASELINE.json config (f_renderer_amd.scenes.build_config), at the reduced and at the full size: depth bits,
triangle ids, RGBA8, the setup records (spi, spf, rhw, varyings; emission order) and the frame counters.

The reference itself cannot run here (Rust; no toolchain) and ships no fixtures, so these digests do NOT pin the
oracle to the reference ("parity unpinned").  What they do: freeze the oracle, so that an edit
of the oracle can no longer move both sides of every parity test silently -- tests/test_golden_frames.py checks the
C oracle, the NumPy oracle (reduced sizes) and the HIP path (-m gpu) against this one file.

  python tests/golden/make_frames.py            # regenerate (only after a REVIEWED change of the oracle)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames.json")


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def setup_digest(spi, spf, rhw, ctx, K):
    """Digest of a setup list given its fields as arrays [n,3,2] i32, [n,3,2] f32, [n,3] f32, [n,3,>=K] f32."""
    return sha(np.asarray(spi, np.int32), np.asarray(spf, np.float32), np.asarray(rhw, np.float32),
               np.asarray(ctx, np.float32)[:, :, :K])


def oracle_entry(cfg):
    from oracle import cref
    from f_renderer_amd import scenes
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    vs, ps = getattr(cref, "VS_" + cfg["vs"]), getattr(cref, "PS_" + cfg["ps"])
    kw = {}
    if cfg["cam"]:
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        kw = dict(view=cref.set_look_at(eye, at, up), proj=cref.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    if cfg["tex"] is not None:
        kw["tex"] = cref.Texture(cfg["tex"])
    u = cref.make_uniforms(flat_color=cfg["flat_color"], **kw)
    f = cref.Frame(W, H)
    f.clear()
    f.draw(mesh, vs, ps, u)
    oc = f.counters.as_dict()
    K = cref.vs_num_varyings(vs)
    setup = cref.geometry_batch(W, H, mesh, vs, u, cap=int(oc["tris_setup"]) + 16)
    assert setup.shape[0] == oc["tris_setup"]
    return dict(width=W, height=H, triangles=int(mesh.shape[0]), vs=cfg["vs"], ps=cfg["ps"], varyings=K,
                tris_setup=oc["tris_setup"], frag_covered=oc["frag_covered"], frag_zpass=oc["frag_zpass"], frag_nan=oc["frag_nan"],
                sha256_depth=sha(f.depth), sha256_tri_id=sha(f.tri_id), sha256_rgba8=sha(f.color),
                sha256_setup=setup_digest(setup["spi"], setup["spf"], setup["rhw"], setup["ctx"], K))


def main():
    from f_renderer_amd import scenes
    out = {}
    for name in scenes.CONFIG_NAMES:
        out[name] = {}
        for size in ("reduced", "full"):
            out[name][size] = oracle_entry(scenes.build_config(name, reduced=size == "reduced"))
            print(name, size, out[name][size]["tris_setup"], out[name][size]["frag_covered"], flush=True)
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")


if __name__ == "__main__":
    main()
