"""Alternate kernel paths kept as fallbacks must produce the same bits as the defaults: FRR_GEOM=scan
(k_scan_blocks between count and emit, the path of meshes with more than 8192 count blocks)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geom", ["scan", "default"])
def test_geometry_paths_agree_with_oracle(oracle, geom, monkeypatch):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    if geom != "default":
        monkeypatch.setenv("FRR_GEOM", geom)
    W, H, n = 400, 300, 70000          # many 256-triangle blocks, dropped (w == 0) and clipped triangles
    tris = scenes.random_clip_triangles(n, W, H, seed=9, spread=1.2)
    tris[::97, 1, 3] = 0.0             # w == 0 on one vertex: triangle dropped (renderer.rs:117-119)
    r = fr.Renderer(W, H)
    f = oracle.Frame(W, H)
    r.clear()
    f.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), keep_setup=True)
    g = r.setup_triangles()
    assert g.shape[0] == setup.shape[0]
    np.testing.assert_array_equal(g["spi"], setup["spi"])
    np.testing.assert_array_equal(g["spf"].view(np.uint32), setup["spf"].view(np.uint32))
    np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup["rhw"].view(np.uint32))
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    assert r.stats()["tris_setup"] == f.counters.tris_setup


@pytest.mark.parametrize("geom", ["scan", "default"])
def test_repeated_draws_per_draw_state_resets(oracle, monkeypatch, geom):
    """Group sums, the clipped-triangle list and the binning counters alternate slots per draw."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    if geom != "default":
        monkeypatch.setenv("FRR_GEOM", geom)
    W, H = 256, 144
    a = scenes.random_clip_triangles(9000, W, H, seed=12)
    b = scenes.random_clip_triangles(3000, W, H, seed=13, spread=1.3)
    r = fr.Renderer(W, H)
    f = oracle.Frame(W, H)
    ma, mb = r.upload_mesh(a, fr.VS_CLIP), r.upload_mesh(b, fr.VS_CLIP)
    u = oracle.make_uniforms()
    for _ in range(2):
        r.clear()
        f.clear()
        f.counters = oracle.Counters()
        base = 0
        for mesh, arr in ((ma, a), (mb, b), (ma, a)):
            r.draw(mesh, fr.PS_DEPTH)
            f.draw(arr, oracle.VS_CLIP, oracle.PS_DEPTH, u, tri_id_base=base)
            base = int(f.counters.tris_setup)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
