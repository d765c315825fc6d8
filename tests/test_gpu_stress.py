"""Randomised small frames against the oracle: odd framebuffer sizes (partial tiles, single-tile and
one-pixel frames), clip-heavy and negative-w triangle soups, two draws per frame, then the same frame on a
2- or 3-rank partition (both layouts; dense-owned geometry) stitched back together."""
import numpy as np
import pytest
from .conftest import owned_pixel_rows

pytestmark = pytest.mark.gpu

CASES = [
    # W, H, n, spread, w_jitter, seed
    (1, 1, 50, 1.0, 0.1, 1), (31, 33, 400, 1.2, 0.1, 2), (32, 32, 900, 1.0, 0.1, 3), (33, 31, 700, 1.5, 0.8, 4),
    (97, 65, 3000, 1.1, 0.1, 5), (257, 129, 5000, 1.3, 0.9, 6), (640, 360, 20000, 1.05, 0.1, 7),
    (500, 37, 4000, 2.0, 1.5, 8), (64, 1000, 6000, 1.2, 0.3, 9), (1280, 96, 9000, 1.1, 0.2, 10),
]


@pytest.mark.parametrize("W,H,n,spread,wj,seed", CASES)
def test_random_frame_and_partition(oracle, W, H, n, spread, wj, seed):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    a = scenes.random_clip_triangles(n, W, H, seed=seed, spread=spread, w_jitter=wj)
    b = scenes.random_clip_triangles(max(n // 3, 1), W, H, seed=seed + 100, spread=spread, w_jitter=wj)
    u = oracle.make_uniforms()
    f = oracle.Frame(W, H)
    f.clear((9, 8, 7, 6), 0.0)
    f.draw(a, oracle.VS_CLIP, oracle.PS_DEPTH, u)
    f.draw(b, oracle.VS_CLIP, oracle.PS_DEPTH, u, tri_id_base=int(f.counters.tris_setup))
    if f.counters.frag_nan:
        pytest.skip("scene produces NaN rhw (unsupported: sticky in the reference)")

    def render(part=None, blocked=False):
        r = fr.Renderer(W, H)
        if part:
            r.set_partition(part[0], part[1], blocked=blocked)
        r.clear((9, 8, 7, 6), 0.0)
        r.draw(r.upload_mesh(a, fr.VS_CLIP), fr.PS_DEPTH)
        r.draw(r.upload_mesh(b, fr.VS_CLIP), fr.PS_DEPTH)
        c, d, t = r.readback()
        st = r.stats()
        r.close()
        return c, d, t, st

    c, d, t, st = render()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    assert st["frag_covered"] == f.counters.frag_covered and st["tris_setup"] == f.counters.tris_setup
    G = 2 + seed % 2
    rows = np.arange(H) // 32
    for blocked in (False, True):
        acc_t = np.full_like(t, 0xFFFFFFFF)
        acc_d = np.zeros_like(d)
        cov = 0
        for rank in range(G):
            _, dr, tr, sr = render((rank, G), blocked)
            own = np.repeat(owned_pixel_rows(H, rank, G, blocked), W)
            acc_t[own] = tr[own]
            acc_d[own] = dr[own]
            cov += sr["frag_covered"]
            assert sr["tris_setup"] == st["tris_setup"]
        np.testing.assert_array_equal(acc_t, t)
        np.testing.assert_array_equal(acc_d.view(np.uint32), d.view(np.uint32))
        assert cov == st["frag_covered"]


WINDOWS = [
    # W, H, (x0, x1), (y0, y1), ranks, blocked, seed     (x0 = 0 whenever y0 > 0 spans several rows: the reference's
    (300, 200, (0, 300), (0, 200), 3, True, 1),        #  depth stride x1 is only self-consistent then)
    (300, 200, (0, 260), (17, 190), 2, False, 2),
    (300, 200, (0, 300), (40, 140), 4, True, 3),
    (256, 256, (0, 200), (0, 256), 2, True, 4),
    (640, 200, (0, 640), (5, 133), 5, False, 5),
]


@pytest.mark.parametrize("W,H,wr,hr,G,blocked,seed", WINDOWS)
def test_coloured_sub_window_on_a_partition(oracle, W, H, wr, hr, G, blocked, seed):
    """Interpolated vertex colours (K = 3) in a sub-window (the clear is settled by k_clear, not fused), on a
    G-rank partition whose tile rows are window-local: every pixel is owned by exactly one rank and the union
    is the oracle's frame (ids, depth, RGBA8)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    n = 5000
    clip = scenes.random_clip_triangles(n, W, H, seed=50 + seed, spread=1.15, w_jitter=0.3)
    col = scenes.splitmix_u01(7 + seed, n * 9).reshape(n, 3, 3).astype(np.float32)
    tris = np.concatenate([clip, col], axis=2)
    f = oracle.Frame(W, H)
    f.clear((5, 6, 7, 8), 0.0)
    f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms(), window=(wr[0], wr[1], hr[0], hr[1]))
    if f.counters.frag_nan:
        pytest.skip("NaN rhw")
    acc_t = np.full(W * H, 0xFFFFFFFF, np.uint32)
    acc_c = np.zeros((H, W, 4), np.uint8)
    acc_c[:] = (5, 6, 7, 8)
    acc_d = np.zeros(W * H, np.float32)
    for rank in range(G):
        r = fr.Renderer(W, H)
        r.set_partition(rank, G, blocked=blocked)
        r.clear((5, 6, 7, 8), 0.0)
        r.draw(r.upload_mesh(tris, fr.VS_CLIP_COLOR), fr.PS_COLOR, wr, hr)
        c, d, t = r.readback()
        drawn = t != 0xFFFFFFFF
        assert not (drawn & (acc_t != 0xFFFFFFFF)).any()
        acc_t[drawn] = t[drawn]
        acc_d[drawn] = d[drawn]
        # colour lives at (cx - x0, cy - y0) with the framebuffer's stride, depth/id at (cy - y0) * x1 + (cx - x0):
        # compare colour through the oracle's own image, pixel by pixel where this rank changed it
        changed = (c != np.array((5, 6, 7, 8), np.uint8)).any(axis=2)
        acc_c[changed] = c[changed]
        r.close()
    np.testing.assert_array_equal(acc_t, f.tri_id)
    np.testing.assert_array_equal(acc_d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(acc_c, f.color)
