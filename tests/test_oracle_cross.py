"""The two independent CPU restatements (C: oracle/frr_oracle.c, NumPy-f32: oracle/oracle_np.py) must
agree bit for bit -- the mitigation for "parity unpinned" (the Rust reference cannot be run here and
ships no fixtures).  Small scenes: the NumPy one loops in Python."""
import numpy as np
import pytest

from oracle import oracle_np as onp


def _np_frame(W, H):
    color = np.zeros((H, W, 4), np.uint8)
    color[...] = (30, 30, 30, 255)
    return color, np.zeros(W * H, np.float32), np.full(W * H, 0xFFFFFFFF, np.uint32)


def _compare(oracle, f, color, depth, tid, setup_c, setup_np, K, check_color=True):
    assert len(setup_np) == setup_c.shape[0]
    for i, tri in enumerate(setup_np):
        for k in range(3):
            assert tuple(setup_c[i, k]["spi"]) == tuple(tri[k]["spi"])
            assert np.array(tri[k]["spf"], np.float32).view(np.uint32).tolist() == setup_c[i, k]["spf"].view(np.uint32).tolist()
            assert np.float32(tri[k]["rhw"]).view(np.uint32) == setup_c[i, k]["rhw"].view(np.uint32)
            if K:
                np.testing.assert_array_equal(tri[k]["ctx"].view(np.uint32), setup_c[i, k]["ctx"][:K].view(np.uint32))
    np.testing.assert_array_equal(tid, f.tri_id)
    np.testing.assert_array_equal(depth.view(np.uint32), f.depth.view(np.uint32))
    if check_color:
        np.testing.assert_array_equal(color, f.color)


@pytest.mark.parametrize("seed,spread,wj", [(1, 0.95, 0.1), (2, 1.3, 0.5), (3, 1.6, 1.5)])
def test_random_clip_triangles(oracle, seed, spread, wj):
    from f_renderer_amd import scenes
    W, H, n = 64, 48, 250
    tris = scenes.random_clip_triangles(n, W * 4, H * 4, seed=seed, spread=spread, w_jitter=wj)  # big relative to the frame
    f = oracle.Frame(W, H)
    f.clear()
    setup_c = f.draw(tris, oracle.VS_CLIP, oracle.PS_FLAT, oracle.make_uniforms(flat_color=(0.5, 0.25, 1.0, 1.0)), keep_setup=True)
    if f.counters.frag_nan:
        pytest.skip("NaN rhw in this scene")
    color, depth, tid = _np_frame(W, H)
    setup_np, cov = onp.draw(W, H, tris, onp.VS_CLIP, onp.PS_FLAT, onp.Uniforms(flat_color=(0.5, 0.25, 1.0, 1.0)), color, depth, tid)
    assert cov == f.counters.frag_covered
    _compare(oracle, f, color, depth, tid, setup_c, setup_np, 0)


def test_vertex_colors(oracle):
    from f_renderer_amd import scenes
    W, H, n = 80, 60, 200
    clip = scenes.random_clip_triangles(n, W * 3, H * 3, seed=7, spread=1.2)
    col = scenes.splitmix_u01(5, n * 9).reshape(n, 3, 3).astype(np.float32)
    tris = np.concatenate([clip, col], axis=2)
    f = oracle.Frame(W, H)
    f.clear()
    setup_c = f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms(), keep_setup=True)
    color, depth, tid = _np_frame(W, H)
    setup_np, _ = onp.draw(W, H, tris, onp.VS_CLIP_COLOR, onp.PS_COLOR, onp.Uniforms(), color, depth, tid)
    _compare(oracle, f, color, depth, tid, setup_c, setup_np, 3)


def _camera(oracle, W, H):
    from f_renderer_amd import scenes
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    vc, pc = oracle.set_look_at(eye, at, up), oracle.set_perspective(fovy, aspect, zn, zf)
    vn, pn = onp.set_look_at(eye, at, up), onp.set_perspective(fovy, aspect, zn, zf)
    np.testing.assert_array_equal(vc.view(np.uint32), vn.view(np.uint32))
    np.testing.assert_array_equal(pc.view(np.uint32), pn.view(np.uint32))
    return eye, vc, pc


def test_torus_gouraud(oracle):
    from f_renderer_amd import scenes
    W, H = 96, 54
    mesh = scenes.torus(nu=14, nv=10)
    eye, view, proj = _camera(oracle, W, H)
    f = oracle.Frame(W, H)
    f.clear()
    setup_c = f.draw(mesh, oracle.VS_GOURAUD, oracle.PS_COLOR, oracle.make_uniforms(view=view, proj=proj, view_pos=eye), keep_setup=True)
    color, depth, tid = _np_frame(W, H)
    setup_np, _ = onp.draw(W, H, mesh, onp.VS_GOURAUD, onp.PS_COLOR, onp.Uniforms(view=view, proj=proj, view_pos=eye), color, depth, tid)
    _compare(oracle, f, color, depth, tid, setup_c, setup_np, 3)


@pytest.mark.parametrize("ps", ["phong", "blinn"])
def test_sphere_textured(oracle, ps):
    from f_renderer_amd import scenes
    W, H = 96, 54
    mesh = scenes.displaced_sphere(n=12)
    tex = scenes.checker_texture(32, 4)
    eye, view, proj = _camera(oracle, W, H)
    f = oracle.Frame(W, H)
    f.clear()
    ps_c = oracle.PS_PHONG if ps == "phong" else oracle.PS_BLINN
    ps_n = onp.PS_PHONG if ps == "phong" else onp.PS_BLINN
    setup_c = f.draw(mesh, oracle.VS_PHONG, ps_c, oracle.make_uniforms(view=view, proj=proj, view_pos=eye, tex=oracle.Texture(tex)), keep_setup=True)
    color, depth, tid = _np_frame(W, H)
    setup_np, _ = onp.draw(W, H, mesh, onp.VS_PHONG, ps_n, onp.Uniforms(view=view, proj=proj, view_pos=eye, tex=tex), color, depth, tid)
    _compare(oracle, f, color, depth, tid, setup_c, setup_np, 8)


def test_window_quirk(oracle):
    """width_range/height_range sub-window: local colour addressing, depth stride = x1 (renderer.rs:362)."""
    from f_renderer_amd import scenes
    W, H = 64, 48
    tris = scenes.random_clip_triangles(150, W * 3, H * 3, seed=11, spread=1.1)
    f = oracle.Frame(W, H)
    f.clear()
    win = (8, 56, 6, 40)
    f.draw(tris, oracle.VS_CLIP, oracle.PS_FLAT, oracle.make_uniforms(), window=win)
    color, depth, tid = _np_frame(W, H)
    onp.draw(W, H, tris, onp.VS_CLIP, onp.PS_FLAT, onp.Uniforms(), color, depth, tid, window=win)
    np.testing.assert_array_equal(tid, f.tri_id)
    np.testing.assert_array_equal(depth.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(color, f.color)
