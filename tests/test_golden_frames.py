"""tests/golden/frames.json freezes the oracle: SHA-256 of depth bits, triangle ids, RGBA8, setup records and the
frame counters of every BASELINE.json config, reduced and full size (tests/golden/make_frames.py wrote it from the C
oracle).  Checked here against (a) the C oracle, (b) the independent NumPy oracle at the reduced sizes, (c) the HIP
path (-m gpu).  The digests do not pin the oracle to the Rust reference (nothing here can: "parity unpinned",
DESIGN.md section 2); they make a silent edit of the oracle fail instead of moving both sides of the parity tests."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_frames", os.path.join(HERE, "golden", "make_frames.py"))
make_frames = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_frames)

with open(os.path.join(HERE, "golden", "frames.json")) as _fh:
    GOLDEN = json.load(_fh)

from f_renderer_amd import scenes  # noqa: E402

CASES = [(n, s) for n in scenes.CONFIG_NAMES for s in ("reduced", "full")]
DIGESTS = ("sha256_depth", "sha256_tri_id", "sha256_rgba8", "sha256_setup")
COUNTS = ("tris_setup", "frag_covered", "frag_zpass", "frag_nan", "triangles", "width", "height")


def test_golden_file_covers_every_config():
    assert sorted(GOLDEN) == sorted(scenes.CONFIG_NAMES)
    for name in GOLDEN:
        assert sorted(GOLDEN[name]) == ["full", "reduced"]
        for e in GOLDEN[name].values():
            assert all(len(e[k]) == 64 for k in DIGESTS) and e["frag_nan"] == 0


@pytest.mark.parametrize("name,size", CASES)
def test_c_oracle_matches_golden(oracle, name, size):
    e = make_frames.oracle_entry(scenes.build_config(name, reduced=size == "reduced"))
    g = GOLDEN[name][size]
    for k in COUNTS + DIGESTS:
        assert e[k] == g[k], (name, size, k)


@pytest.mark.parametrize("name", scenes.CONFIG_NAMES)
def test_numpy_oracle_matches_golden(name):
    from oracle import oracle_np as onp
    cfg = scenes.build_config(name, reduced=True)
    g = GOLDEN[name]["reduced"]
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    vs, ps = getattr(onp, "VS_" + cfg["vs"]), getattr(onp, "PS_" + cfg["ps"])
    kw = {}
    if cfg["cam"]:
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        kw = dict(view=onp.set_look_at(eye, at, up), proj=onp.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    if cfg["tex"] is not None:
        kw["tex"] = cfg["tex"]
    u = onp.Uniforms(flat_color=cfg["flat_color"], **kw)
    color = np.zeros((H, W, 4), np.uint8)
    color[...] = (30, 30, 30, 255)
    depth = np.zeros(W * H, np.float32)
    tid = np.full(W * H, 0xFFFFFFFF, np.uint32)
    setup, cov = onp.draw(W, H, mesh, vs, ps, u, color, depth, tid)
    K = onp.VS_K[vs]
    assert len(setup) == g["tris_setup"] and cov == g["frag_covered"]
    n = len(setup)
    spi = np.array([[v["spi"] for v in t] for t in setup], np.int32).reshape(n, 3, 2)
    spf = np.array([[v["spf"] for v in t] for t in setup], np.float32).reshape(n, 3, 2)
    rhw = np.array([[v["rhw"] for v in t] for t in setup], np.float32).reshape(n, 3)
    ctx = np.array([[np.asarray(v["ctx"], np.float32)[:K] if K else np.zeros(0, np.float32) for v in t] for t in setup], np.float32).reshape(n, 3, K)
    assert make_frames.setup_digest(spi, spf, rhw, ctx, K) == g["sha256_setup"]
    assert make_frames.sha(depth) == g["sha256_depth"]
    assert make_frames.sha(tid) == g["sha256_tri_id"]
    assert make_frames.sha(color) == g["sha256_rgba8"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", CASES)
def test_hip_path_matches_golden(name, size):
    """The HIP path against the committed digests alone (no oracle in the loop)."""
    import f_renderer_amd as fr
    cfg = scenes.build_config(name, reduced=size == "reduced")
    g = GOLDEN[name][size]
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    vs, ps = getattr(fr, "VS_" + cfg["vs"]), getattr(fr, "PS_" + cfg["ps"])
    r = fr.Renderer(W, H)
    kw = {}
    if cfg["cam"]:
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        kw = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    if cfg["tex"] is not None:
        r.set_texture(0, cfg["tex"])
        kw["texture_slot"] = 0
    r.set_uniforms(flat_color=cfg["flat_color"], **kw)
    m = r.upload_mesh(mesh, vs)
    r.clear()
    r.draw(m, ps)
    c, d, t = r.readback()
    st = r.stats()
    assert st["tris_setup"] == g["tris_setup"] and st["frag_covered"] == g["frag_covered"] and st["frag_nan"] == 0
    assert make_frames.sha(d) == g["sha256_depth"]
    assert make_frames.sha(t) == g["sha256_tri_id"]
    if cfg["ps"] != "DEPTH":   # depth-only draws leave the colour target at the clear colour, like the oracle's
        assert make_frames.sha(c) == g["sha256_rgba8"]
    else:
        assert make_frames.sha(c) == g["sha256_rgba8"]
    s = r.setup_triangles()
    K = fr.lib().frr_vs_num_varyings(vs)
    assert make_frames.setup_digest(s["spi"], s["spf"], s["rhw"], s["ctx"], K) == g["sha256_setup"]
    # the early-z build (statistic off) must produce the same image
    r.set_count_fragments(False)
    r.clear()
    r.draw(m, ps)
    c2, d2, t2 = r.readback()
    assert make_frames.sha(d2) == g["sha256_depth"] and make_frames.sha(t2) == g["sha256_tri_id"] and make_frames.sha(c2) == g["sha256_rgba8"]
