"""GPU parity tests: the HIP path (through the C ABI, include/frr.h) against the CPU oracle on the
same seeded inputs.  Bar: bit-exact coverage, triangle ids, depth bits, RGBA8 bytes and setup
records (integer AND float fields) -- the HIP path reproduces the reference's fp32 operation
order, so no tolerance is needed or used.
"""
import numpy as np
import pytest
from .conftest import assert_depth_equal, owned_pixel_rows

pytestmark = pytest.mark.gpu


def _mk(oracle, W, H):
    import f_renderer_amd as fr
    return fr.Renderer(W, H), oracle.Frame(W, H)


def _assert_frame_equal(r, f, check_color=True, stats=True):
    c, d, t = r.readback()
    do = f.depth
    np.testing.assert_array_equal(t, f.tri_id, err_msg="triangle-id buffer differs")
    assert_depth_equal(d, do)
    if check_color:
        np.testing.assert_array_equal(c, f.color, err_msg="RGBA8 differs")
    if stats:
        s = r.stats()
        oc = f.counters.as_dict()
        assert s["frag_nan"] == oc["frag_nan"]
        assert s["tris_in"] == oc["tris_in"]
        assert s["tris_setup"] == oc["tris_setup"]
        assert s["frag_covered"] == oc["frag_covered"]


def _assert_setup_equal(r, setup_o, K):
    g = r.setup_triangles()
    assert g.shape[0] == setup_o.shape[0]
    np.testing.assert_array_equal(g["spi"], setup_o["spi"])
    np.testing.assert_array_equal(g["spf"].view(np.uint32), setup_o["spf"].view(np.uint32))
    np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup_o["rhw"].view(np.uint32))
    if K:
        np.testing.assert_array_equal(g["ctx"][..., :K].view(np.uint32), setup_o["ctx"][..., :K].view(np.uint32))


def test_kat_b1_single_triangle(oracle):
    """BASELINE config 1 / SURVEY Appendix B.1: 32,640 covered pixels, flat shading, z cleared to 0."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    r, f = _mk(oracle, 512, 512)
    tri = scenes.single_triangle()
    r.set_uniforms(flat_color=(1.0, 0.25, 0.5, 1.0))
    u = oracle.make_uniforms(flat_color=(1.0, 0.25, 0.5, 1.0))
    r.clear((30, 30, 30, 255), 0.0)
    f.clear((30, 30, 30, 255), 0.0)
    m = r.upload_mesh(tri, fr.VS_CLIP)
    r.draw(m, fr.PS_FLAT)
    f.draw(tri, oracle.VS_CLIP, oracle.PS_FLAT, u)
    _assert_frame_equal(r, f)
    assert r.stats()["frag_covered"] == 32640
    c, _, t = r.readback()
    cov = t.reshape(512, 512) != 0xFFFFFFFF
    assert cov.sum() == 32640 and cov[129, 256] and not cov[128].any() and not cov[384].any()
    assert tuple(c[200, 256]) == (255, 63, 127, 255)


def test_rgb_triangle_interpolated(oracle):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    r, f = _mk(oracle, 512, 512)
    tri = scenes.single_triangle_rgb()
    u = oracle.make_uniforms()
    r.clear((0, 0, 0, 0), 0.0)
    f.clear((0, 0, 0, 0), 0.0)
    r.draw(r.upload_mesh(tri, fr.VS_CLIP_COLOR), fr.PS_COLOR)
    f.draw(tri, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, u)
    _assert_frame_equal(r, f)


@pytest.mark.parametrize("W,H,n,spread,wj,seed", [
    (256, 256, 5000, 0.95, 0.1, 1),
    (333, 211, 20000, 1.15, 0.1, 2),      # ragged size, many clipped triangles
    (320, 180, 1500, 1.3, 1.5, 3),        # negative / tiny w: the quirky clipper's kept outside vertices,
                                          # screen-filling fans, saturated spi and wrapping edge functions
    (64, 64, 3000, 2.0, 0.5, 4),          # most triangles partly or wholly off screen
])
def test_random_triangles_depth_and_setup(oracle, W, H, n, spread, wj, seed):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    r, f = _mk(oracle, W, H)
    tris = scenes.random_clip_triangles(n, W, H, seed=seed, spread=spread, w_jitter=wj)
    u = oracle.make_uniforms()
    r.clear()
    f.clear()
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.draw(m, fr.PS_DEPTH)
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, u, keep_setup=True)
    _assert_setup_equal(r, setup, 0)
    _assert_frame_equal(r, f)


def test_random_triangles_color(oracle):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H, n = 320, 200, 8000
    r, f = _mk(oracle, W, H)
    clip = scenes.random_clip_triangles(n, W, H, seed=11, spread=1.1)
    col = scenes.splitmix_u01(99, n * 9).reshape(n, 3, 3).astype(np.float32)
    tris = np.concatenate([clip, col], axis=2)
    u = oracle.make_uniforms()
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP_COLOR), fr.PS_COLOR)
    setup = f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, u, keep_setup=True)
    _assert_setup_equal(r, setup, 3)
    _assert_frame_equal(r, f)


def _camera_uniforms(oracle, r, W, H, tex=None, slot=0):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    view_g, proj_g = fr.set_look_at(eye, at, up), fr.set_perspective(fovy, aspect, zn, zf)
    view_o, proj_o = oracle.set_look_at(eye, at, up), oracle.set_perspective(fovy, aspect, zn, zf)
    np.testing.assert_array_equal(view_g.view(np.uint32), view_o.view(np.uint32))
    np.testing.assert_array_equal(proj_g.view(np.uint32), proj_o.view(np.uint32))
    otex = None
    if tex is not None:
        r.set_texture(slot, tex)
        otex = oracle.Texture(tex)
    r.set_uniforms(view=view_g, proj=proj_g, view_pos=eye, texture_slot=slot)
    return oracle.make_uniforms(view=view_o, proj=proj_o, view_pos=eye, tex=otex)


def test_torus_gouraud(oracle):
    """BASELINE config 2 (teapot-class, 6,272 triangles), at 640x360 for CPU time."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 640, 360
    r, f = _mk(oracle, W, H)
    mesh = scenes.torus()
    u = _camera_uniforms(oracle, r, W, H)
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(mesh, fr.VS_GOURAUD), fr.PS_COLOR)
    setup = f.draw(mesh, oracle.VS_GOURAUD, oracle.PS_COLOR, u, keep_setup=True)
    _assert_setup_equal(r, setup, 3)
    _assert_frame_equal(r, f)


@pytest.mark.parametrize("ps", ["phong", "blinn"])
def test_sphere_textured(oracle, ps):
    """BASELINE config 3 (bunny-class) at reduced tessellation/size: K=8 perspective-correct
    varyings, bilinear texture, reference Phong (phong.rs:133-154) and the Blinn variant."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 480, 270
    r, f = _mk(oracle, W, H)
    mesh = scenes.displaced_sphere(n=64)
    tex = scenes.checker_texture(256, 16)
    u = _camera_uniforms(oracle, r, W, H, tex)
    ps_g = fr.PS_PHONG if ps == "phong" else fr.PS_BLINN
    ps_o = oracle.PS_PHONG if ps == "phong" else oracle.PS_BLINN
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(mesh, fr.VS_PHONG), ps_g)
    setup = f.draw(mesh, oracle.VS_PHONG, ps_o, u, keep_setup=True)
    _assert_setup_equal(r, setup, 8)
    _assert_frame_equal(r, f)


def test_sheets_clipped_textured(oracle):
    """BASELINE config 5 class (stacked sheets, large + small triangles, outer layers clipped)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 384, 216
    r, f = _mk(oracle, W, H)
    mesh = scenes.layered_sheets(gx=40, gy=16, layers=5)
    tex = scenes.checker_texture(128, 8)
    u = _camera_uniforms(oracle, r, W, H, tex)
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(mesh, fr.VS_PHONG), fr.PS_BLINN)
    setup = f.draw(mesh, oracle.VS_PHONG, oracle.PS_BLINN, u, keep_setup=True)
    _assert_setup_equal(r, setup, 8)
    _assert_frame_equal(r, f)


def test_multi_draw_ties_and_ids(oracle):
    """Two draws into one frame; the second repeats the first's triangles, so every fragment ties
    and the LATER triangle must own the pixel (renderer.rs:363: `rhw < depth` rejects, equal passes)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 200, 120
    r, f = _mk(oracle, W, H)
    a = scenes.random_clip_triangles(1500, W, H, seed=21)
    u1 = oracle.make_uniforms(flat_color=(1, 0, 0, 1))
    u2 = oracle.make_uniforms(flat_color=(0, 1, 0, 1))
    r.clear()
    f.clear()
    m = r.upload_mesh(a, fr.VS_CLIP)
    r.set_uniforms(flat_color=(1, 0, 0, 1))
    r.draw(m, fr.PS_FLAT)
    r.set_uniforms(flat_color=(0, 1, 0, 1))
    r.draw(m, fr.PS_FLAT)
    f.draw(a, oracle.VS_CLIP, oracle.PS_FLAT, u1)
    n1 = int(f.counters.tris_setup)
    f.draw(a, oracle.VS_CLIP, oracle.PS_FLAT, u2, tri_id_base=n1)
    _assert_frame_equal(r, f)
    _, _, t = r.readback()
    drawn = t != 0xFFFFFFFF
    assert drawn.any() and (t[drawn] >= n1).all()


def test_window_subrange(oracle):
    """width_range/height_range sub-window with the reference's local addressing and its depth
    stride quirk (renderer.rs:323,326,362,381)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 160
    r, f = _mk(oracle, W, H)
    tris = scenes.random_clip_triangles(4000, W, H, seed=31, spread=1.05)
    u = oracle.make_uniforms(flat_color=(0.2, 0.4, 0.6, 1))
    r.set_uniforms(flat_color=(0.2, 0.4, 0.6, 1))
    r.clear()
    f.clear()
    wr, hr = (40, 200), (30, 140)
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_FLAT, wr, hr)
    f.draw(tris, oracle.VS_CLIP, oracle.PS_FLAT, u, window=(wr[0], wr[1], hr[0], hr[1]))
    _assert_frame_equal(r, f)


def test_window_errors(oracle):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    r = fr.Renderer(64, 64)
    m = r.upload_mesh(scenes.single_triangle(), fr.VS_CLIP)
    r.clear()
    with pytest.raises(fr.FrrError) as e:
        r.draw(m, fr.PS_DEPTH, (10, 5), (0, 64))  # clamp(min > max) panics in the reference
    assert e.value.code == fr.FRR_ERR_INVALID
    with pytest.raises(fr.FrrError):
        r.draw(m, fr.PS_COLOR)  # K mismatch
    r.draw(m, fr.PS_DEPTH, (0, 0), (0, 0))  # empty window is fine
    r.sync()


@pytest.mark.parametrize("blocked", [False, True])
def test_tile_partition_stitch(oracle, blocked):
    """Screen-tile partition (multi-GPU scheme): rank r of G renders tile rows ty % G == r (interleaved) or
    the contiguous rows [r*k, (r+1)*k), k = ceil(rows/G) (blocked); the union of the G partial images must
    be byte-identical to the 1-GPU image."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H, G = 300, 200, 3
    tris = scenes.random_clip_triangles(6000, W, H, seed=41, spread=1.1)
    full = fr.Renderer(W, H)
    full.clear()
    full.draw(full.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    c0, d0, t0 = full.readback()
    acc_d = np.zeros_like(d0)
    acc_t = np.full_like(t0, 0xFFFFFFFF)
    rows = np.arange(H) // 32
    for rank in range(G):
        r = fr.Renderer(W, H)
        r.set_partition(rank, G, blocked=blocked)
        r.clear()
        r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        assert r.stats()["tris_setup"] == full.stats()["tris_setup"]     # the reference's count, on every rank
        own = np.repeat(owned_pixel_rows(H, rank, G, blocked), W)
        assert (t[~own] == 0xFFFFFFFF).all()
        acc_d[own] = d[own]
        acc_t[own] = t[own]
        r.close()
    np.testing.assert_array_equal(acc_t, t0)
    np.testing.assert_array_equal(acc_d.view(np.uint32), d0.view(np.uint32))


def test_determinism_repeated_runs(oracle):
    """The z resolution uses LDS atomics and unordered bins; the image must not depend on that."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 256
    tris = scenes.random_clip_triangles(30000, W, H, seed=51)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    ref = None
    for _ in range(3):
        r.clear()
        r.draw(m, fr.PS_DEPTH)
        _, d, t = r.readback()
        if ref is None:
            ref = (d.copy(), t.copy())
        else:
            np.testing.assert_array_equal(t, ref[1])
            np.testing.assert_array_equal(d.view(np.uint32), ref[0].view(np.uint32))


def test_device_atan2f_matches_glibc(oracle):
    """The device atan2f (fdlibm port) vs this box's glibc atan2f, which Rust's f32::atan2 calls
    (renderer.rs:208-209)."""
    import ctypes
    import f_renderer_amd as fr
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(7)
    n = 200000
    y = rng.standard_normal(n).astype(np.float32) * np.float32(10) ** rng.integers(-6, 6, n).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32) * np.float32(10) ** rng.integers(-6, 6, n).astype(np.float32)
    y[:8] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-40, 3.0]
    x[:8] = [-1.0, -1.0, 0.0, -0.0, np.inf, -np.inf, -1e-40, 1.0]
    r = fr.Renderer(32, 32)
    got = r.debug_atan2f(y, x)
    exp = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y[:20000], x[:20000])], np.float32)
    np.testing.assert_array_equal(got[:20000].view(np.uint32), exp.view(np.uint32))
    host = np.array([fr.lib().frr_host_atan2f(float(a), float(b)) for a, b in zip(y[:20000], x[:20000])], np.float32)
    np.testing.assert_array_equal(host.view(np.uint32), exp.view(np.uint32))


def test_wave_scan_dpp():
    """The DPP wave64 inclusive scan used by the span rasterizer, against numpy."""
    import f_renderer_amd as fr
    r = fr.Renderer(32, 32)
    rng = np.random.default_rng(3)
    for _ in range(4):
        v = rng.integers(0, 33, 64).astype(np.uint32)
        np.testing.assert_array_equal(r.debug_scan64(v), np.cumsum(v).astype(np.uint32))
    np.testing.assert_array_equal(r.debug_scan64(np.ones(64, np.uint32)), np.arange(1, 65, dtype=np.uint32))


@pytest.mark.parametrize("mode", ["sweep"])
def test_sweep_kernel_still_exact(oracle, mode, monkeypatch):
    """FRR_RASTER=sweep selects the brute-force tile kernel (the span kernel's fallback form)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_RASTER", mode)
    W, H = 333, 211
    r, f = _mk(oracle, W, H)
    tris = scenes.random_clip_triangles(20000, W, H, seed=2, spread=1.15)
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    _assert_frame_equal(r, f)


def test_fast_reciprocal_is_ieee_exact_for_every_float():
    """recip_exact (v_rcp_f32 + one exact-residual Newton step, IEEE division outside [2^-64, 2^65)) is
    what the fragment loop uses for `1.0 / s` (renderer.rs:356), and rsqrt_exact (v_sqrt_f32 + the neighbour test with
    exact residuals, then the same reciprocal) is the shaders' `normalize` (phong.rs:136-141): both against the
    compiler's IEEE operations for all 2^32 bit patterns on the device."""
    import f_renderer_amd as fr
    r = fr.Renderer(64, 64)
    total = 0
    for lo, hi in ((0, 0x40000000), (0x40000000, 0x80000000), (0x80000000, 0xC0000000), (0xC0000000, 0xFFFFFFFF)):
        n, first = r.debug_rcp_check(lo, hi)
        assert n == 0, f"{n} mismatches, first at bit pattern {first:#x}"
        total += n
    assert total == 0
