"""CPU tests of the oracle (oracle/frr_oracle.c) against hand-derivable known-answer vectors and
properties.  The reference has NO tests or fixtures for this path (examples/src/lib.rs:1-8 is
`2+2==4`), so these integer KATs -- derived from renderer.rs:285-341 alone, see SURVEY.md
Appendix B and tests/golden/kat_integer.json -- are what pins the restatement ("parity unpinned"
otherwise)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _vtx(oracle, spi, spf=None, rhw=1.0, pos=None):
    v = np.zeros(3, oracle.VERTEX_DTYPE)
    for i in range(3):
        v[i]["spi"] = spi[i]
        v[i]["spf"] = spf[i] if spf is not None else spi[i]
        v[i]["rhw"] = rhw
        # NDC pos only feeds the orientation test (renderer.rs:300-312): y up => flip
        v[i]["pos"] = pos[i] if pos is not None else (spi[i][0], -spi[i][1], 0.0, 1.0)
    return v


def test_kat_b1_config1_triangle(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "kat_integer.json")))["b1"]
    u = oracle.make_uniforms(flat_color=(1, 0, 0, 1))
    tri = np.array(kat["clip"], np.float32)
    out = oracle.geometry_processing(512, 512, tri, oracle.VS_CLIP, u)
    assert out.shape == (1, 3)
    assert out["spi"][0].tolist() == kat["sorted_spi"]           # centroid-angle sort: [v2, v0, v1]
    assert out["spf"][0].tolist() == [[float(a), float(b)] for a, b in kat["sorted_spi"]]
    f = oracle.Frame(512, 512)
    f.clear((0, 0, 0, 0), 0.0)
    f.draw(tri, oracle.VS_CLIP, oracle.PS_FLAT, u)
    cov = f.tri_id.reshape(512, 512) != 0xFFFFFFFF
    assert int(cov.sum()) == kat["covered"] == f.counters.frag_covered
    ys, xs = np.nonzero(cov)
    assert [int(xs[0]), int(ys[0])] == kat["first"] and [int(xs[-1]), int(ys[-1])] == kat["last"]
    assert np.nonzero(cov[129])[0].tolist() == [256]
    r383 = np.nonzero(cov[383])[0]
    assert (r383[0], r383[-1]) == (129, 383)
    assert not cov[128].any() and not cov[384].any()            # exclusive bbox upper bound
    assert (f.depth[cov.reshape(-1)] == 1.0).all()              # all rhw_i = 1
    assert f.color[200, 256].tolist() == [255, 0, 0, 255]


def test_kat_b2_watertight_quad(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "kat_integer.json")))["b2"]
    u = oracle.make_uniforms()
    f = oracle.Frame(256, 256)
    f.clear((0, 0, 0, 0), 0.0)
    counts = []
    hits = np.zeros((256, 256), np.int32)
    for k, t in enumerate(kat["tris"]):
        g = oracle.Frame(256, 256)
        g.clear((0, 0, 0, 0), 0.0)
        rc = g.rasterization((0, 256), (0, 256), _vtx(oracle, t), oracle.PS_FLAT, 0, u, tri_id=k)
        assert rc == 0
        counts.append(int(g.counters.frag_covered))
        hits += (g.tri_id.reshape(256, 256) != 0xFFFFFFFF)
    assert counts == kat["covered"]
    assert hits[100:200, 100:200].min() == 1 and hits.max() == 1 and int(hits.sum()) == 10000


def test_orientation_swap_needed(oracle):
    """B.1: without the orientation swap (renderer.rs:309-312) the same vertices cover 0 pixels."""
    u = oracle.make_uniforms()
    spi = [(256, 128), (128, 384), (384, 384)]
    f = oracle.Frame(512, 512)
    f.clear((0, 0, 0, 0), 0.0)
    f.rasterization((0, 512), (0, 512), _vtx(oracle, spi), oracle.PS_FLAT, 0, u)
    assert f.counters.frag_covered == 32640
    g = oracle.Frame(512, 512)
    g.clear((0, 0, 0, 0), 0.0)
    # lie about NDC so that normal.z <= 0 and no swap happens
    g.rasterization((0, 512), (0, 512), _vtx(oracle, spi, pos=[(0, 0, 0, 1)] * 3), oracle.PS_FLAT, 0, u)
    assert g.counters.frag_covered == 0


def test_fan_emission_order(oracle):
    """A clipped triangle: list = [intersections..., v0, v1, v2], sorted by angle, fanned as
    [0,k-1,k] for k=n-1..4, then [0,2,3], [0,1,2] (renderer.rs:245-266)."""
    u = oracle.make_uniforms()
    tri = np.array([[-0.5, -0.5, 0.5, 1.0], [1.5, -0.5, 0.5, 1.0], [0.0, 0.5, 0.5, 1.0]], np.float32)  # v1 beyond x=w
    out = oracle.geometry_processing(100, 100, tri, oracle.VS_CLIP, u)
    # edges (0,1) and (1,2) cross X_RIGHT -> 2 intersections + 3 originals = 5 vertices -> 3 triangles
    assert out.shape == (3, 3)
    v = {tuple(x) for x in out["spi"].reshape(-1, 2).tolist()}
    assert len(v) == 5
    # all three triangles share the fan apex (first sorted vertex)
    assert (out["spi"][:, 0] == out["spi"][0, 0]).all()
    # last emitted is [s0, s1, s2], the one before is [s0, s2, s3]
    assert out["spi"][2, 2].tolist() == out["spi"][1, 1].tolist()


def test_w_zero_drops_triangle(oracle):
    u = oracle.make_uniforms()
    tri = np.array([[0, 0, 0, 0.0], [1, 0, 0.5, 1], [0, 1, 0.5, 1]], np.float32)
    assert oracle.geometry_processing(64, 64, tri, oracle.VS_CLIP, u).shape[0] == 0
    tri[0, 3] = -0.0
    assert oracle.geometry_processing(64, 64, tri, oracle.VS_CLIP, u).shape[0] == 0


def test_z_rule_later_wins_ties(oracle):
    """renderer.rs:363: `rhw < depth` rejects; equal passes, so the later triangle owns the pixel."""
    u = oracle.make_uniforms()
    spi = [(10, 10), (10, 50), (50, 50)]
    f = oracle.Frame(64, 64)
    f.clear((0, 0, 0, 0), 0.0)
    f.rasterization((0, 64), (0, 64), _vtx(oracle, spi, rhw=0.5), oracle.PS_FLAT, 0, u, tri_id=1)
    f.rasterization((0, 64), (0, 64), _vtx(oracle, spi, rhw=0.5), oracle.PS_FLAT, 0, u, tri_id=2)
    f.rasterization((0, 64), (0, 64), _vtx(oracle, spi, rhw=0.25), oracle.PS_FLAT, 0, u, tri_id=3)  # farther: rejected
    ids = f.tri_id[f.tri_id != 0xFFFFFFFF]
    assert ids.size > 0 and (ids == 2).all()


def test_quantise_and_sample(oracle):
    v = np.array([1.0, 0.999, -0.5, np.nan], np.float32)
    out = np.zeros(4, np.uint8)
    import ctypes as C
    oracle.lib().o_vec4_to_u8(v.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert out.tolist() == [255, 254, 0, 0]                      # truncation, clamp, NaN -> 0
    tex = np.zeros((4, 4, 4), np.uint8)
    tex[..., 0] = np.arange(16).reshape(4, 4) * 17
    t = oracle.Texture(tex)
    rc, px = t.sample_2d(0.0, 0.0)
    assert rc == 0 and px[0] == 0.0
    rc, px = t.sample_2d(0.375, 0.0)                             # x = 1.5 -> halfway texel 1 and 2
    assert rc == 0 and px[0] == np.float32((np.float32(17 / 255.0 * 0.5) * 1.0) + 0.0 + np.float32(np.float32(34 / 255.0) * 0.5) * 1.0)
    wide = oracle.Texture(np.zeros((2, 4, 4), np.uint8))         # height < width: y clamp uses width -> OOB
    rc, _ = wide.sample_2d(0.0, 0.99)
    assert rc == -1


def test_window_invariance_full_vs_windows(oracle):
    """Tile-window property: rendering each window (x0,x1)x(y0,y1) with x0 = 0 into its own frame
    equals the crop of the full render (windows are how the reference exposes tiling)."""
    from f_renderer_amd import scenes
    W, H = 96, 64
    tris = scenes.random_clip_triangles(1500, W, H, seed=77, spread=1.1)
    u = oracle.make_uniforms()
    full = oracle.Frame(W, H)
    full.clear()
    full.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, u)
    fd = full.depth.reshape(H, W)
    for (y0, y1) in [(0, 32), (32, 64)]:
        part = oracle.Frame(W, H)
        part.clear()
        # geometry uses the full viewport; only the raster window changes
        setup = oracle.geometry_batch(W, H, tris, oracle.VS_CLIP, u)
        for i in range(setup.shape[0]):
            part.rasterization((0, W), (y0, y1), setup[i], oracle.PS_DEPTH, 0, u, tri_id=i)
        pd = part.depth.reshape(H, W)[: y1 - y0]
        np.testing.assert_array_equal(pd.view(np.uint32), fd[y0:y1].view(np.uint32))


def test_banded_multithread_frame_equals_the_single_thread_frame(oracle):
    """oracle.cref.draw_banded (bench.py's all-core CPU figure) splits the rows over threads through the
    reference's sub-window argument; the stitched frame is the single-thread frame."""
    from f_renderer_amd import scenes
    W, H = 200, 150
    tris = scenes.random_clip_triangles(3000, W, H, seed=4, spread=1.2)
    u = oracle.make_uniforms()
    f = oracle.Frame(W, H)
    f.clear((1, 2, 3, 4), 0.0)
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, u)
    for threads in (1, 3, 8):
        c, d, t, cov, _ = oracle.draw_banded(W, H, tris, oracle.VS_CLIP, oracle.PS_DEPTH, u, (1, 2, 3, 4), 0.0, threads=threads)
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        np.testing.assert_array_equal(c, f.color)
        assert cov == f.counters.frag_covered
