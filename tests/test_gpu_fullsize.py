"""BASELINE.json's full-size workloads through the C ABI: the headline frame (1,000,000 random clip-space
triangles, 1920x1080, depth-only) bit for bit against the oracle (it needs ~1 s of CPU for that frame),
and the size-independent properties the domain offers at that size: the 8-rank tile partition stitches
to the single-GPU image and its fragment counts add up; splitting the draw call in two leaves every
pixel unchanged (emission indices are frame-global); early-z on/off and repeated runs are identical."""
import numpy as np
import pytest
from .conftest import owned_pixel_rows

pytestmark = pytest.mark.gpu

W, H, N = 1920, 1080, 1_000_000


@pytest.fixture(scope="module")
def headline():
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    tris = scenes.random_clip_triangles(N, W, H)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.set_count_fragments(True)
    r.clear()
    r.draw(m, fr.PS_DEPTH)
    c, d, t = r.readback()
    st = r.stats()
    yield dict(tris=tris, r=r, mesh=m, depth=d, ids=t, stats=st)
    r.close()


def test_headline_frame_bit_exact_against_oracle(oracle, headline):
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(headline["tris"], oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    oc = f.counters.as_dict()
    assert oc["frag_nan"] == 0 and headline["stats"]["frag_nan"] == 0
    np.testing.assert_array_equal(headline["ids"], f.tri_id)
    np.testing.assert_array_equal(headline["depth"].view(np.uint32), f.depth.view(np.uint32))
    assert headline["stats"]["frag_covered"] == oc["frag_covered"] == 60_715_175
    assert headline["stats"]["tris_setup"] == oc["tris_setup"] == 1_000_052


def test_headline_early_z_and_repeat_are_identical(headline):
    import f_renderer_amd as fr
    r = headline["r"]
    r.set_count_fragments(False)          # whole-triangle early-z on (the timed configuration)
    for _ in range(8):                    # waves race freely on the LDS keys: every run must give the same bits
        r.clear()
        r.draw(headline["mesh"], fr.PS_DEPTH)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, headline["ids"])
        np.testing.assert_array_equal(d.view(np.uint32), headline["depth"].view(np.uint32))
    r.set_count_fragments(True)


@pytest.mark.parametrize("blocked", [False, True])
def test_headline_partition_of_8_stitches_and_counts_add_up(headline, blocked):
    import f_renderer_amd as fr
    G = 8
    rows = np.arange(H) // 32
    acc_t = np.full(W * H, 0xFFFFFFFF, np.uint32)
    acc_d = np.zeros(W * H, np.float32)
    covered = 0
    for rank in range(G):
        r = fr.Renderer(W, H)
        r.set_partition(rank, G, blocked=blocked)
        r.set_count_fragments(True)
        r.clear()
        r.draw(r.upload_mesh(headline["tris"], fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        covered += r.stats()["frag_covered"]
        own = np.repeat(owned_pixel_rows(H, rank, G, blocked), W)
        acc_t[own] = t[own]
        acc_d[own] = d[own]
        r.close()
    np.testing.assert_array_equal(acc_t, headline["ids"])
    np.testing.assert_array_equal(acc_d.view(np.uint32), headline["depth"].view(np.uint32))
    assert covered == headline["stats"]["frag_covered"]           # every covered fragment belongs to exactly one rank


def test_headline_split_into_two_draws_is_the_same_frame(headline):
    import f_renderer_amd as fr
    tris = headline["tris"]
    r = fr.Renderer(W, H)
    a, b = r.upload_mesh(tris[: N // 3], fr.VS_CLIP), r.upload_mesh(tris[N // 3:], fr.VS_CLIP)
    r.clear()
    r.draw(a, fr.PS_DEPTH)
    r.draw(b, fr.PS_DEPTH)
    _, d, t = r.readback()
    st = r.stats()
    np.testing.assert_array_equal(t, headline["ids"])             # ids are frame-global emission indices
    np.testing.assert_array_equal(d.view(np.uint32), headline["depth"].view(np.uint32))
    assert st["draws"] == 2 and st["tris_setup"] == headline["stats"]["tris_setup"]
    r.close()


def _camera(oracle, fr, scenes, Wc, Hc):
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(Wc, Hc)
    g = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    o = dict(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    return g, o


@pytest.mark.parametrize("name", ["cfg2_torus_gouraud", "cfg3_sphere_phong", "cfg3b_sphere_blinn", "cfg4_random_4096", "cfg5_sheets_4k_blinn"])
def test_baseline_config_full_size_bit_exact(oracle, name):
    """BASELINE.json configs 2-5 at their full sizes (same scenes as tools/run_configs.py): depth bits,
    triangle ids, RGBA8 and the fragment / setup counts against the oracle."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    tex = scenes.checker_texture(1024, 32)
    cfg = {
        "cfg2_torus_gouraud": (1920, 1080, scenes.torus, "GOURAUD", "COLOR", True, None),
        "cfg3_sphere_phong": (1920, 1080, scenes.displaced_sphere, "PHONG", "PHONG", True, tex),
        "cfg3b_sphere_blinn": (1920, 1080, scenes.displaced_sphere, "PHONG", "BLINN", True, tex),
        "cfg4_random_4096": (4096, 4096, lambda: scenes.random_clip_triangles(1_000_000, 4096, 4096), "CLIP", "DEPTH", False, None),
        "cfg5_sheets_4k_blinn": (3840, 2160, scenes.layered_sheets, "PHONG", "BLINN", True, tex),
    }[name]
    Wc, Hc, mk, vs, ps, cam, texture = cfg
    mesh = mk()
    r = fr.Renderer(Wc, Hc)
    gkw, okw = ({}, {})
    if cam:
        gkw, okw = _camera(oracle, fr, scenes, Wc, Hc)
    if texture is not None:
        r.set_texture(0, texture)
        gkw["texture_slot"] = 0
        okw["tex"] = oracle.Texture(texture)
    r.set_uniforms(flat_color=(1.0, 0.5, 0.25, 1.0), **gkw)
    r.clear()
    r.draw(r.upload_mesh(mesh, getattr(fr, "VS_" + vs)), getattr(fr, "PS_" + ps))
    c, d, t = r.readback()
    st = r.stats()
    f = oracle.Frame(Wc, Hc)
    f.clear()
    f.draw(mesh, getattr(oracle, "VS_" + vs), getattr(oracle, "PS_" + ps), oracle.make_uniforms(flat_color=(1.0, 0.5, 0.25, 1.0), **okw))
    oc = f.counters.as_dict()
    assert oc["frag_nan"] == 0 and st["frag_nan"] == 0
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    if ps != "DEPTH":
        np.testing.assert_array_equal(c, f.color)
    assert st["frag_covered"] == oc["frag_covered"] and st["tris_setup"] == oc["tris_setup"]
    r.close()


def test_three_million_triangles_take_the_large_mesh_paths(oracle):
    """3,000,000 triangles at 1920x1080: more than 8192 count blocks (k_scan_blocks between count and emit),
    bin chunks of ~11.7K triangles (several rounds per workgroup, regions larger than the LDS staging) and
    tiles with ~2,600 records (the sorted path's overflow loops) -- bit for bit against the oracle."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    n = 3_000_000
    tris = scenes.random_clip_triangles(n, W, H, seed=31)
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    _, d, t = r.readback()
    st = r.stats()
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    oc = f.counters.as_dict()
    assert oc["frag_nan"] == 0 and st["frag_nan"] == 0
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    assert st["frag_covered"] == oc["frag_covered"] and st["tris_setup"] == oc["tris_setup"]
    r.close()


@pytest.mark.parametrize("blocked", [False, True])
def test_clip_heavy_textured_scene_partitioned_equals_single_gpu(blocked):
    """BASELINE config 5 (250,000-triangle sheets, a third of them through or past the frustum, textured
    Blinn-Phong, 3840x2160) on a 4-rank partition: dense-owned geometry with many clipped triangles; the
    stitched colour, depth and id images must equal the single-context render."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    Wc, Hc, G = 3840, 2160, 4
    mesh = scenes.layered_sheets()
    tex = scenes.checker_texture(1024, 32)
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(Wc, Hc)
    kw = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)

    def render(rank=None):
        r = fr.Renderer(Wc, Hc)
        if rank is not None:
            r.set_partition(rank, G, blocked=blocked)
        r.set_texture(0, tex)
        r.set_uniforms(**kw)
        r.clear()
        r.draw(r.upload_mesh(mesh, fr.VS_PHONG), fr.PS_BLINN)
        out = r.readback()
        st = r.stats()
        r.close()
        return out, st

    (c0, d0, t0), st0 = render()
    rows = np.arange(Hc) // 32
    acc_c, acc_d, acc_t = np.zeros_like(c0), np.zeros_like(d0), np.zeros_like(t0)
    for rank in range(G):
        (c, d, t), st = render(rank)
        assert st["tris_setup"] == st0["tris_setup"]
        own_rows = owned_pixel_rows(Hc, rank, G, blocked)
        own = np.repeat(own_rows, Wc)
        acc_d[own] = d[own]
        acc_t[own] = t[own]
        acc_c[own_rows] = c[own_rows]
    np.testing.assert_array_equal(acc_t, t0)
    np.testing.assert_array_equal(acc_d.view(np.uint32), d0.view(np.uint32))
    np.testing.assert_array_equal(acc_c, c0)
