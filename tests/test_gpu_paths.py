"""Kernel paths and per-draw state: many geometry blocks with dropped and clipped inputs, repeated and mixed draws,
every shape of the tile kernel, the fan region's capacity growth."""
import numpy as np
import pytest
from .conftest import assert_depth_equal

pytestmark = pytest.mark.gpu


def test_geometry_paths_agree_with_oracle(oracle):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
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


def test_repeated_draws_per_draw_state_resets(oracle):
    """The fan cursor and the binning counters alternate slots per draw."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
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


@pytest.mark.parametrize("nw,occ", [(3, 6), (4, 6), (4, 8), (6, 6), (8, 6), (16, 4)])
@pytest.mark.parametrize("scene", ["depth_heavy", "phong"])
def test_tile_kernel_shapes_agree_with_oracle(oracle, monkeypatch, nw, occ, scene):
    """Every (waves per tile, register budget) build of k_raster_span the host may pick (span_shape in
    frr_api.hip), forced through FRR_RASTER_NW / FRR_RASTER_OCC: same bits as the oracle.  The depth scene has
    tiles above and below the direct-path limit (256 records) and more than 192 binning chunks' worth of
    triangles; the Phong scene runs the textured resolve (K = 8)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_RASTER_NW", str(nw))
    monkeypatch.setenv("FRR_RASTER_OCC", str(occ))
    if scene == "depth_heavy":
        W, H, n = 200, 136, 300000
        tris = scenes.random_clip_triangles(n, W, H, seed=31, spread=1.05)
        r = fr.Renderer(W, H)
        f = oracle.Frame(W, H)
        r.set_count_fragments(False)
        r.clear()
        f.clear()
        r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    else:
        W, H = 320, 200
        mesh = scenes.displaced_sphere(n=40)
        tex = scenes.checker_texture(64, 8)
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        r = fr.Renderer(W, H)
        r.set_texture(0, tex)
        r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
        r.clear()
        r.draw(r.upload_mesh(mesh, fr.VS_PHONG), fr.PS_PHONG)
        f = oracle.Frame(W, H)
        f.clear()
        u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                                 view_pos=eye, tex=oracle.Texture(tex))
        f.draw(mesh, oracle.VS_PHONG, oracle.PS_PHONG, u)
        c, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        np.testing.assert_array_equal(c, f.color)


def test_mixed_geometry_paths_across_draws(oracle):
    """The per-draw slot (fan cursor) alternates per draw whichever path a draw takes: a small mesh, 2.2M triangles
    (more than 8192 geometry blocks) and the empty mesh must each leave the other slot clean for the draw after them
    -- small -> 2.2M triangles -> small, over two frames, then small -> empty -> small with clipped triangles."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 144
    small = scenes.random_clip_triangles(5000, W, H, seed=41, spread=1.3)          # many clipped triangles
    big = scenes.random_clip_triangles(2_200_000, 8 * W, 8 * H, seed=42)            # 8593 count blocks; tiny triangles
    empty = np.zeros((0, 3, 4), np.float32)
    r = fr.Renderer(W, H)
    f = oracle.Frame(W, H)
    u = oracle.make_uniforms()
    ms, mb, me = r.upload_mesh(small, fr.VS_CLIP), r.upload_mesh(big, fr.VS_CLIP), r.upload_mesh(empty, fr.VS_CLIP)
    for seq in (((ms, small), (mb, big), (ms, small)), ((ms, small), (mb, big), (ms, small)),
                ((ms, small), (me, empty), (ms, small)), ((me, empty), (ms, small), (me, empty), (ms, small))):
        r.clear()
        f.clear()
        f.counters = oracle.Counters()
        base = 0
        for mesh, arr in seq:
            r.draw(mesh, fr.PS_DEPTH)
            if arr.shape[0]:
                f.draw(arr, oracle.VS_CLIP, oracle.PS_DEPTH, u, tri_id_base=base)
            base = int(f.counters.tris_setup)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        st = r.stats()
        assert st["tris_setup"] == f.counters.tris_setup and st["replays"] == 0


def test_filtered_setup_list_is_not_reused(oracle):
    """frr_draw on a partitioned ctx filters the setup list by the rank's tile rows of its window; frr_raster with
    another window / partition and frr_readback_setup must refuse it (FRR_ERR_INVALID) instead of silently
    dropping triangles, and the unfiltered frr_geometry path serves any number of windows (renderer.rs:269-271)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 160
    tris = scenes.random_clip_triangles(4000, W, H, seed=43)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.set_partition(1, 2)
    r.clear()
    r.draw(m, fr.PS_DEPTH)
    with pytest.raises(fr.FrrError) as e:
        r.rasterization((0, W), (0, H // 2), fr.PS_DEPTH)
    assert e.value.code == fr.FRR_ERR_INVALID
    with pytest.raises(fr.FrrError):
        r.setup_triangles()
    r.set_partition(0, 2)
    with pytest.raises(fr.FrrError):
        r.rasterization((0, W), (0, H), fr.PS_DEPTH)
    # the unfiltered list: two windows of one geometry on a partitioned ctx, both ranks stitched == oracle
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    full_d = np.zeros(W * H, np.float32)
    full_t = np.full(W * H, 0xFFFFFFFF, np.uint32)
    for rank in range(2):
        r.set_partition(rank, 2)
        r.clear()
        n = r.geometry_processing(m, count=True)
        assert n == f.counters.tris_setup
        assert r.setup_triangles().shape[0] == n
        r.rasterization((0, W), (0, H), fr.PS_DEPTH)
        _, d, t = r.readback()
        rows = np.arange(H) // 32 % 2 == rank
        mask = np.repeat(rows, W)
        full_d[mask] = d[mask]
        full_t[mask] = t[mask]
    np.testing.assert_array_equal(full_t, f.tri_id)
    np.testing.assert_array_equal(full_d.view(np.uint32), f.depth.view(np.uint32))


def test_fan_capacity_overflow_is_replayed_inside_the_library(oracle):
    """A mesh whose clipped inputs need more fan slots than the first guess (max(one per input + 4096, min(19 per
    input, 2^20))): the geometry pass fails on the device, the library grows the fan space and replays the draw itself."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H, n = 64, 48, 300000
    tris = scenes.random_clip_triangles(n, 2 * W, 2 * H, seed=51, spread=0.2)
    tris[..., :2] *= 40.0            # every triangle straddles the frustum: ~4 fan triangles per input
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    if f.counters.frag_nan:
        pytest.skip("NaN rhw in this scene")
    assert f.counters.tris_setup > (1 << 20) + 100000
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.clear()
    r.draw(m, fr.PS_DEPTH)
    r.sync()                                              # no error: the replay happened in here
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    st = r.stats()
    assert st["tris_setup"] == f.counters.tris_setup and st["replays"] >= 1


@pytest.mark.parametrize("clip_queue", [0, 1])
def test_small_fan_space_is_grown_by_replays(oracle, monkeypatch, clip_queue):
    """Option fan_capacity forces a tiny fan space on a clip-heavy mesh (cheap to run): setup records, ids and depth are the
    oracle's after the library's own replays, also when frr_geometry is asked for its count and in a second draw of the frame."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_FAN_CAP", "64")
    monkeypatch.setenv("FRR_CLIP_QUEUE", str(clip_queue))
    W, H = 200, 150
    tris = scenes.random_clip_triangles(6000, W, H, seed=8, spread=1.3, w_jitter=0.5)
    small = scenes.random_clip_triangles(50, W, H, seed=9, spread=0.5)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(small, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), keep_setup=True, tri_id_base=int(f.counters.tris_setup))
    if f.counters.frag_nan:
        pytest.skip("NaN rhw")
    assert setup.shape[0] > 6000 + 64
    r = fr.Renderer(W, H)
    m0, m = r.upload_mesh(small, fr.VS_CLIP), r.upload_mesh(tris, fr.VS_CLIP)
    r.clear()
    r.draw(m0, fr.PS_DEPTH)
    n = r.geometry_processing(m, count=True)             # the count is the reference's, replay or not
    assert n == setup.shape[0]
    r.rasterization((0, W), (0, H), fr.PS_DEPTH)
    g = r.setup_triangles()
    np.testing.assert_array_equal(g["spi"], setup["spi"])
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert_depth_equal(d, f.depth)
    st = r.stats()
    assert st["replays"] >= 1 and st["tris_setup"] == f.counters.tris_setup and st["draws"] == 2


@pytest.mark.parametrize("scene", ["clip_heavy", "phong", "many_rounds"])
def test_draw_of_clipped_textured_and_many_round_meshes(oracle, scene):
    """frr_draw (geometry + segmented binning + tile kernel) twice in a row -- the per-draw slots alternate -- on clipped
    fans, a textured mesh (K = 8 varyings) and a mesh of 600k triangles: setup records, ids, depth and colour are the
    oracle's bits."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    if scene == "phong":
        W, H = 320, 200
        mesh = scenes.displaced_sphere(n=60)
        tex = scenes.checker_texture(64, 8)
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        r = fr.Renderer(W, H)
        r.set_texture(0, tex)
        r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
        f = oracle.Frame(W, H)
        u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                                 view_pos=eye, tex=oracle.Texture(tex))
        vs, ps, ovs, ops = fr.VS_PHONG, fr.PS_BLINN, oracle.VS_PHONG, oracle.PS_BLINN
    else:
        W, H = (300, 170) if scene == "clip_heavy" else (640, 360)
        n = 30000 if scene == "clip_heavy" else 600000
        mesh = scenes.random_clip_triangles(n, W if scene == "clip_heavy" else 4 * W, H if scene == "clip_heavy" else 4 * H,
                                            seed=61, spread=1.25 if scene == "clip_heavy" else 1.0)
        r = fr.Renderer(W, H)
        f = oracle.Frame(W, H)
        u = oracle.make_uniforms()
        vs, ps, ovs, ops = fr.VS_CLIP, fr.PS_DEPTH, oracle.VS_CLIP, oracle.PS_DEPTH
    m = r.upload_mesh(mesh, vs)
    for _ in range(2):                      # twice: the per-draw slots alternate
        r.clear()
        r.draw(m, ps)
    f.clear()
    setup = f.draw(mesh, ovs, ops, u, keep_setup=scene != "many_rounds")
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    if scene == "phong":
        np.testing.assert_array_equal(c, f.color)
    st = r.stats()
    assert st["tris_setup"] == f.counters.tris_setup and st["replays"] == 0
    if scene != "many_rounds":
        g = r.setup_triangles()
        assert g.shape[0] == setup.shape[0]
        np.testing.assert_array_equal(g["spi"], setup["spi"])
        np.testing.assert_array_equal(g["spf"].view(np.uint32), setup["spf"].view(np.uint32))
        np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup["rhw"].view(np.uint32))


@pytest.mark.parametrize("mode", ["0", "1", "auto"])
def test_clip_queue_gives_the_same_setup_and_frame(oracle, monkeypatch, mode):
    """Clipped inputs beyond four per block can be expanded by a second launch (k_geom_clip, option clip_queue) instead
    of by their geometry block -- the tail of meshes that clip along their index order.  Forced on, forced off and in the
    automatic mode (which switches on after a counter read-back has shown a block with many clipped inputs) the setup
    records, ids, depth and colour are the oracle's bits: a scene where every second triangle clips, and the layered
    sheets of config 5 (whole runs of consecutive clipped triangles) through the textured shader."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    if mode != "auto":
        monkeypatch.setenv("FRR_CLIP_QUEUE", mode)
    W, H = 300, 170
    tris = scenes.random_clip_triangles(30000, W, H, seed=61, spread=1.25)
    f = oracle.Frame(W, H)
    f.clear()
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), keep_setup=True)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    for _ in range(3):                      # (auto: the first frame's read-back turns the queue on for the next)
        r.clear()
        r.draw(m, fr.PS_DEPTH)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        g = r.setup_triangles()
        assert g.shape[0] == setup.shape[0]
        np.testing.assert_array_equal(g["spi"], setup["spi"])
        np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup["rhw"].view(np.uint32))
    cfg = scenes.build_config("cfg5", reduced=True)
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    r = fr.Renderer(W, H)
    r.set_texture(0, cfg["tex"])
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    f = oracle.Frame(W, H)
    u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                             view_pos=eye, tex=oracle.Texture(cfg["tex"]))
    f.clear()
    setup = f.draw(mesh, oracle.VS_PHONG, oracle.PS_BLINN, u, keep_setup=True)
    m = r.upload_mesh(mesh, fr.VS_PHONG)
    for _ in range(2):
        r.clear()
        r.draw(m, fr.PS_BLINN)
        c, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        np.testing.assert_array_equal(c, f.color)
        g = r.setup_triangles()
        assert g.shape[0] == setup.shape[0]
        np.testing.assert_array_equal(g["ctx"][..., :8].view(np.uint32), setup["ctx"][..., :8].view(np.uint32))


@pytest.mark.parametrize("world", [1, 3])
def test_binning_chunks_of_more_than_64_blocks(oracle, world):
    """A binning workgroup walks its geometry blocks 64 at a time (bin_walk_blocks: the per-block counts of the dense
    binning entries are scanned by one wave).  Two chunks over 60,000 triangles (118 blocks each; what a mesh of more than
    16 M triangles does with 255 chunks), whole window and a 3-way partition stitched together: ids, depth and the setup
    counts are the oracle's."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 400, 300
    tris = scenes.random_clip_triangles(60000, W, H, seed=77, spread=1.2)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    got_t = np.full(W * H, 0xFFFFFFFF, np.uint32)
    got_d = np.zeros(W * H, np.float32)
    for rank in range(world):
        r = fr.Renderer(W, H)
        r.set_option("bin_chunks", 2)
        if world > 1:
            r.set_partition(rank, world, blocked=True)
        for _ in range(2):
            r.clear()
            r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        assert r.stats()["tris_setup"] == f.counters.tris_setup
        for y0, y1 in (r.owned_rows((0, H)) if world > 1 else [(0, H)]):
            got_t[y0 * W:y1 * W] = t[y0 * W:y1 * W]
            got_d[y0 * W:y1 * W] = d[y0 * W:y1 * W]
        r.close()
    np.testing.assert_array_equal(got_t, f.tri_id)
    np.testing.assert_array_equal(got_d.view(np.uint32), f.depth.view(np.uint32))
