"""frr_clear is deferred and normally performed by the tile kernel of the next full-window draw
(frr_api.hip: settle / fused_clear).  Every ordering of clear / geometry / raster / readback must
still behave like the reference's `frame_buffer.fill(..); depth_buffer.fill(..)` (phong.rs:316-317)
followed by its draw loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RGBA = (11, 22, 33, 44)


def _scene(W, H, n=3000, seed=5):
    from f_renderer_amd import scenes
    return scenes.random_clip_triangles(n, W, H, seed=seed, spread=1.1)


def _oracle(oracle, tris, W, H, rgba=RGBA, depth=0.0, window=None):
    f = oracle.Frame(W, H)
    f.clear(rgba, depth)
    kw = {"window": window} if window else {}
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), **kw)
    return f


def _same(r, f):
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)


def test_clear_alone_then_readback_and_stats():
    import f_renderer_amd as fr
    W, H = 130, 70
    r = fr.Renderer(W, H)
    r.clear(RGBA, 0.25)
    st = r.stats()
    assert st["tris_in"] == 0 and st["frag_covered"] == 0 and st["draws"] == 0
    c, d, t = r.readback()
    assert (c.reshape(-1, 4) == np.array(RGBA, np.uint8)).all()
    assert (d == np.float32(0.25)).all() and (t == 0xFFFFFFFF).all()


@pytest.mark.parametrize("eager", [False, True])
def test_fused_clear_matches_oracle_including_colour(oracle, monkeypatch, eager):
    import f_renderer_amd as fr
    if eager:
        monkeypatch.setenv("FRR_CLEAR", "eager")
    W, H = 300, 170                                   # partial tiles on both edges, some empty tiles
    tris = _scene(W, H, n=400, seed=6)
    f = _oracle(oracle, tris, W, H)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    for _ in range(2):                                # second frame starts from the first frame's content
        r.clear(RGBA, 0.0)
        r.draw(m, fr.PS_DEPTH)
        _same(r, f)
    assert r.stats()["tris_in"] == len(tris)          # counters were reset by the second frame's first draw


def test_clear_after_draw_without_new_draw(oracle):
    import f_renderer_amd as fr
    W, H = 160, 96
    tris = _scene(W, H)
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    r.clear(RGBA, 0.5)
    c, d, t = r.readback()
    assert (c.reshape(-1, 4) == np.array(RGBA, np.uint8)).all() and (d == np.float32(0.5)).all() and (t == 0xFFFFFFFF).all()
    assert r.stats()["tris_in"] == 0


def test_pending_clear_with_sub_window_draw(oracle):
    import f_renderer_amd as fr
    W, H = 256, 160
    tris = _scene(W, H)
    win = (0, 200, 0, 120)                            # x0 = y0 = 0 keeps the reference's depth stride self-consistent
    f = _oracle(oracle, tris, W, H, window=win)
    r = fr.Renderer(W, H)
    r.clear(RGBA, 0.0)
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH, (win[0], win[1]), (win[2], win[3]))
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(c, f.color)


def test_two_draws_only_first_is_fused(oracle):
    import f_renderer_amd as fr
    W, H = 200, 120
    a, b = _scene(W, H, 2000, 1), _scene(W, H, 2000, 2)
    f = oracle.Frame(W, H)
    f.clear(RGBA, 0.0)
    u = oracle.make_uniforms()
    f.draw(a, oracle.VS_CLIP, oracle.PS_DEPTH, u)
    f.draw(b, oracle.VS_CLIP, oracle.PS_DEPTH, u, tri_id_base=int(f.counters.tris_setup))
    r = fr.Renderer(W, H)
    r.clear(RGBA, 0.0)
    r.draw(r.upload_mesh(a, fr.VS_CLIP), fr.PS_DEPTH)
    r.draw(r.upload_mesh(b, fr.VS_CLIP), fr.PS_DEPTH)
    _same(r, f)
    assert r.stats()["draws"] == 2


def test_geometry_before_clear_draws_nothing(oracle):
    """frr_clear resets n_setup (k_clear always did): geometry -> clear -> raster leaves the clear values."""
    import f_renderer_amd as fr
    W, H = 128, 64
    r = fr.Renderer(W, H)
    r.geometry_processing(r.upload_mesh(_scene(W, H), fr.VS_CLIP))
    r.clear(RGBA, 0.0)
    r.rasterization((0, W), (0, H), fr.PS_DEPTH)
    c, d, t = r.readback()
    assert (t == 0xFFFFFFFF).all() and (c.reshape(-1, 4) == np.array(RGBA, np.uint8)).all()


def test_partitioned_rank_owes_the_other_rows_only_until_somebody_looks(oracle):
    import f_renderer_amd as fr
    W, H, G = 192, 160, 2
    tris = _scene(W, H)
    f = _oracle(oracle, tris, W, H)
    rows = np.arange(H) // 32
    for rank in range(G):
        r = fr.Renderer(W, H)
        r.set_partition(rank, G)
        for _ in range(2):
            r.clear(RGBA, 0.0)
            r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        c, d, t = r.readback()
        own = np.repeat((rows % G) == rank, W)
        np.testing.assert_array_equal(t[own], f.tri_id[own])
        np.testing.assert_array_equal(c.reshape(-1, 4)[own], f.color.reshape(-1, 4)[own])
        assert (t[~own] == 0xFFFFFFFF).all()
        assert (c.reshape(-1, 4)[~own] == np.array(RGBA, np.uint8)).all()
        r.close()
