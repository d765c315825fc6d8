"""Edge cases and the less-travelled code paths of the HIP library, all against the oracle:
capacity overflow + re-issue, global-atomic binning fallback, device-bound meshes and targets,
empty / degenerate inputs, non-square textures, several texture slots."""
import numpy as np
import pytest
from .conftest import assert_depth_equal

pytestmark = pytest.mark.gpu


def _oracle_depth(oracle, tris, W, H):
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    return f


@pytest.mark.parametrize("path", ["segmented", "atomics"])
def test_bin_capacity_overflow_is_replayed_inside_the_library(oracle, monkeypatch, path):
    """Renderer::rasterization cannot fail (renderer.rs:269-384): a (triangle, tile) list that turns out too small is
    grown and the draw replayed by the library itself; the caller sees a correct frame and `replays` in the statistics."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_BIN_CAP", "2000")            # far too small on purpose
    if path == "atomics":
        monkeypatch.setenv("FRR_BIN", "atomics")
    W, H = 256, 192
    tris = scenes.random_clip_triangles(4000, W, H, seed=3, spread=1.0)
    f = _oracle_depth(oracle, tris, W, H)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.clear()
    r.draw(m, fr.PS_DEPTH)
    _, d, t = r.readback()                                # no error, no re-issue by the caller
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    st = r.stats()
    assert st["replays"] >= 1 and st["tris_in"] == 4000 and st["draws"] == 1
    assert st["tris_setup"] == f.counters.tris_setup and st["frag_covered"] == f.counters.frag_covered
    r.clear()
    r.draw(m, fr.PS_DEPTH)                                # the lists have grown: the next frame needs no replay
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert r.stats()["replays"] == 0


def test_replay_in_the_middle_of_a_frame(oracle, monkeypatch):
    """Three draws into one frame, the SECOND needs more bin space than there is: the first draw's results stay, the second
    and third are replayed (the third was cancelled on the device when the second failed) -- ids, depth, colour and the
    frame statistics are those of a frame that never failed.  Also with caller-bound targets that change between frames."""
    import torch
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 320, 240
    n = (300, 9000, 700)
    meshes = []
    for k, nk in enumerate(n):
        clip = scenes.random_clip_triangles(nk, W, H, seed=90 + k, spread=1.1, w_jitter=0.4)
        col = scenes.splitmix_u01(17 + k, nk * 9).reshape(nk, 3, 3).astype(np.float32)
        meshes.append(np.concatenate([clip, col], axis=2))
    f = oracle.Frame(W, H)
    f.clear((9, 8, 7, 6), 0.0)
    for mk in meshes:
        f.draw(mk, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms(), tri_id_base=int(f.counters.tris_setup))
    if f.counters.frag_nan:
        pytest.skip("NaN rhw")
    monkeypatch.setenv("FRR_BIN_CAP", "6000")            # enough for the first and third draw, not for the second
    r = fr.Renderer(W, H)
    ms = [r.upload_mesh(mk, fr.VS_CLIP_COLOR) for mk in meshes]
    sets = [(torch.zeros((H, W), dtype=torch.int32, device="cuda"), torch.zeros((H, W), dtype=torch.float32, device="cuda"),
             torch.zeros((H, W), dtype=torch.int32, device="cuda")) for _ in range(2)]
    for frame in range(3):
        c_, d_, t_ = sets[frame % 2]
        r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
        r.clear((9, 8, 7, 6), 0.0)
        for m in ms:
            r.draw(m, fr.PS_COLOR)
        r.sync()
        torch.cuda.synchronize()
        st = r.stats()
        assert st["replays"] == (1 if frame == 0 else 0), st
        assert st["draws"] == 3 and st["tris_in"] == sum(n) and st["tris_setup"] == f.counters.tris_setup
        assert st["frag_covered"] == f.counters.frag_covered
        np.testing.assert_array_equal(t_.cpu().numpy().view(np.uint32).ravel(), f.tri_id)
        np.testing.assert_array_equal(d_.cpu().numpy().view(np.uint32).ravel(), f.depth.view(np.uint32))
        np.testing.assert_array_equal(c_.cpu().numpy().view(np.uint8).reshape(H, W, 4), f.color)
    r.close()


def test_replay_of_separate_geometry_and_raster_calls(oracle, monkeypatch):
    """frr_geometry once, frr_raster over two windows (renderer.rs:269-271 allows reusing one geometry): the second window's
    binning overflows; only that raster pass is replayed, on the intact setup list."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 256
    tris = scenes.random_clip_triangles(5000, W, H, seed=12, spread=1.0)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), window=(0, W, 0, 64))
    f2 = oracle.Frame(W, H)
    f2.clear()
    f2.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    monkeypatch.setenv("FRR_BIN_CAP", "4000")
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    r.clear()
    r.geometry_processing(m)
    r.rasterization((0, W), (0, 64), fr.PS_DEPTH)        # a quarter of the frame: fits
    _, d, t = r.readback()
    assert r.stats()["replays"] == 0
    np.testing.assert_array_equal(t, f.tri_id)
    r.clear()
    r.geometry_processing(m)
    r.rasterization((0, W), (0, H), fr.PS_DEPTH)         # the whole frame: does not
    _, d, t = r.readback()
    assert r.stats()["replays"] >= 1
    np.testing.assert_array_equal(t, f2.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f2.depth.view(np.uint32))


def test_global_atomic_binning_fallback(oracle, monkeypatch):
    """The path used when a frame has more tiles than fit LDS counters (> 36,864 tiles)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_BIN", "atomics")
    W, H = 300, 200
    tris = scenes.random_clip_triangles(9000, W, H, seed=4, spread=1.25, w_jitter=0.6)   # incl. huge fans
    f = _oracle_depth(oracle, tris, W, H)
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    assert r.stats()["frag_covered"] == f.counters.frag_covered


def test_device_bound_mesh_and_targets(oracle):
    """frr_mesh_bind_device / frr_bind_targets with torch-owned HBM (the bench.py plumbing)."""
    import torch
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 320, 224
    tris = scenes.random_clip_triangles(5000, W, H, seed=5)
    f = _oracle_depth(oracle, tris, W, H)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        r = fr.Renderer(W, H, stream=s.cuda_stream)
        color = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        depth = torch.zeros((H, W), dtype=torch.float32, device="cuda")
        ids = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        r.bind_targets(color.data_ptr(), depth.data_ptr(), ids.data_ptr())
        assert r.target_ptrs() == (color.data_ptr(), depth.data_ptr(), ids.data_ptr())
        dev = torch.from_numpy(tris).cuda()
        m = r.bind_mesh_device(dev.data_ptr(), tris.shape[0], fr.VS_CLIP, keepalive=dev)
        r.clear()
        r.draw(m, fr.PS_DEPTH)
        r.sync()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(depth.cpu().numpy().reshape(-1).view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(ids.cpu().numpy().reshape(-1).view(np.uint32), f.tri_id)


def test_empty_and_degenerate_inputs(oracle):
    import f_renderer_amd as fr
    W, H = 64, 64
    r = fr.Renderer(W, H)
    f = oracle.Frame(W, H)
    r.clear()
    f.clear()
    empty = r.upload_mesh(np.zeros((0, 3, 4), np.float32), fr.VS_CLIP)
    r.draw(empty, fr.PS_DEPTH)                                           # no triangles at all
    degen = np.array([
        [[0.1, 0.1, 0.5, 1], [0.1, 0.1, 0.5, 1], [0.1, 0.1, 0.5, 1]],    # all three vertices equal
        [[-0.5, 0.0, 0.5, 1], [0.0, 0.0, 0.5, 1], [0.5, 0.0, 0.5, 1]],   # collinear
        [[0.2, 0.2, 0.5, 0], [0.3, 0.2, 0.5, 1], [0.2, 0.3, 0.5, 1]],    # w == 0 -> dropped
        [[-0.9, -0.9, 0.5, 1], [0.9, -0.9, 0.5, 1], [0.0, 0.9, 0.5, 1]], # a real one
        [[-0.9, -0.9, 0.5, 1], [0.9, -0.9, 0.5, 1], [0.0, 0.9, 0.5, 1]], # its duplicate: later wins ties
    ], np.float32)
    r.set_uniforms(flat_color=(0.25, 0.5, 0.75, 1.0))
    r.draw(r.upload_mesh(degen, fr.VS_CLIP), fr.PS_FLAT)
    f.draw(degen, oracle.VS_CLIP, oracle.PS_FLAT, oracle.make_uniforms(flat_color=(0.25, 0.5, 0.75, 1.0)))
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    st = r.stats()
    assert st["tris_in"] == 5 and st["tris_setup"] == f.counters.tris_setup == 4 and st["draws"] == 2


def test_non_square_texture_and_slots(oracle):
    """height > width is legal (sample_2d clamps y with width, renderer.rs:523); height < width is
    rejected because the reference would index out of bounds; uniforms.texture_slot = `place`."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 200, 120
    mesh = scenes.displaced_sphere(n=20)
    tall = scenes.checker_texture(64, 8)
    tall = np.concatenate([tall, tall[::-1]], axis=0)                   # 128 x 64
    other = scenes.checker_texture(32, 4)
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    r = fr.Renderer(W, H)
    with pytest.raises(fr.FrrError) as e:
        r.set_texture(0, np.zeros((16, 32, 4), np.uint8))
    assert e.value.code == fr.FRR_ERR_UNSUPPORTED
    r.set_texture(0, other)
    r.set_texture(2, tall)
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=2)
    r.clear()
    r.draw(r.upload_mesh(mesh, fr.VS_PHONG), fr.PS_PHONG)
    f = oracle.Frame(W, H)
    f.clear()
    u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                             view_pos=eye, tex=oracle.Texture(tall))
    f.draw(mesh, oracle.VS_PHONG, oracle.PS_PHONG, u)
    c, d, _ = r.readback()
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    r.set_uniforms(texture_slot=3)                                      # nothing uploaded there
    with pytest.raises(fr.FrrError):
        r.draw(r.upload_mesh(mesh, fr.VS_PHONG), fr.PS_PHONG)


def test_partition_with_window(oracle):
    """tile-row partition combined with a sub-window (tile rows are window-local)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H, G = 256, 192, 2
    tris = scenes.random_clip_triangles(5000, W, H, seed=8, spread=1.1)
    wr, hr = (16, 240), (10, 170)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), window=(wr[0], wr[1], hr[0], hr[1]))
    acc = np.full(W * H, 0xFFFFFFFF, np.uint32)
    for rank in range(G):
        r = fr.Renderer(W, H)
        r.set_partition(rank, G)
        r.clear()
        r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH, wr, hr)
        _, _, t = r.readback()
        drawn = t != 0xFFFFFFFF
        assert not (drawn & (acc != 0xFFFFFFFF)).any()                  # ranks never touch each other's pixels
        acc[drawn] = t[drawn]
        r.close()
    np.testing.assert_array_equal(acc, f.tri_id)


def test_vertex_sort_near_ties_and_axes(oracle):
    """The device decides the centroid-angle order of an unclipped triangle with half-plane / cross
    predicates and falls back to the exact atan2f keys near ties and near the 0 / pi axis
    (k_geom_emit).  Adversarial triangles around those switches must still give the oracle's
    setup records (vertex order, swap, snapped corners) bit for bit."""
    import f_renderer_amd as fr
    W, H = 256, 256
    rng = np.random.default_rng(77)
    tris = []
    deltas = [0.0] + [s * 10.0 ** e for e in range(-8, -1) for s in (1.0, -1.0)]
    for base in np.arange(0.0, 2.0 * np.pi, np.pi / 4.0):
        for d0 in deltas:
            for d1 in deltas:
                c = rng.uniform(-0.5, 0.5, 2)
                r0, r1 = rng.uniform(0.05, 0.3, 2)
                a0, a1 = base + d0, base + d0 + d1
                v0 = c + r0 * np.array([np.cos(a0), np.sin(a0)])
                v1 = c + r1 * np.array([np.cos(a1), np.sin(a1)]) * (1.0 if rng.random() < 0.5 else -1.0)
                v2 = 3.0 * c - v0 - v1                       # keeps the centroid (nearly) at c
                tris.append([[v0[0], v0[1], 0.5, 1.0], [v1[0], v1[1], 0.5, 1.0], [v2[0], v2[1], 0.5, 1.0]])
    # exact axis cases: a vertex exactly level with / above the centroid, repeated vertices, zero area
    for k in range(200):
        c = np.round(rng.uniform(-0.5, 0.5, 2) * 64.0) / 64.0
        h = 2.0 ** -rng.integers(2, 6)
        kind = k % 4
        if kind == 0:
            p = [c + [2 * h, 0], c + [-h, h], c + [-h, -h]]           # d0 on the +x axis exactly
        elif kind == 1:
            p = [c + [-2 * h, 0], c + [h, h], c + [h, -h]]            # d0 on the -x axis exactly
        elif kind == 2:
            p = [c + [h, 0], c + [-h, 0], c + [0, 0]]                 # collinear through the centroid
        else:
            p = [c + [h, h], c + [h, h], c + [-2 * h, -2 * h]]        # repeated vertex
        tris.append([[q[0], q[1], 0.5, 1.0] for q in p])
    tris = np.asarray(tris, dtype=np.float32)
    perm = rng.permutation(len(tris))
    tris = np.ascontiguousarray(tris[perm])
    f = oracle.Frame(W, H)
    f.clear()
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), keep_setup=True)
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    g = r.setup_triangles()
    assert g.shape[0] == setup.shape[0]
    np.testing.assert_array_equal(g["spi"], setup["spi"])
    np.testing.assert_array_equal(g["spf"].view(np.uint32), setup["spf"].view(np.uint32))
    np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup["rhw"].view(np.uint32))
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert_depth_equal(d, f.depth)


@pytest.mark.parametrize("slot", [1, 16])
def test_hot_tiles_use_the_overflow_arena(oracle, monkeypatch, slot):
    """Segmented binning gives every tile a fixed slot for its near-first record copy; tiles with more
    records than the slot allocate from the shared overflow arena.  Force that with a tiny slot."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    monkeypatch.setenv("FRR_ENT_SLOT", str(slot))
    W, H = 320, 200
    tris = scenes.random_clip_triangles(6000, W, H, seed=9, spread=1.1)
    f = _oracle_depth(oracle, tris, W, H)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    for _ in range(3):                                     # consecutive draws alternate the counter slots
        r.clear()
        r.draw(m, fr.PS_DEPTH)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
        assert r.stats()["frag_covered"] == f.counters.frag_covered
        assert r.stats()["bin_entries"] > 0


def test_clustered_clipped_triangles(oracle):
    """Every triangle of the mesh is clipped (a screen-filling sheet reaching past the frustum): the
    clipped-triangle list built by k_geom_count is shared by all waves of k_geom_emit."""
    import f_renderer_amd as fr
    W, H, n = 256, 160, 48
    xs = np.linspace(-1.6, 1.6, n + 1, dtype=np.float32)
    tris = []
    for i in range(n):
        for sgn in (1.0, -1.0):                                   # a strip above and one below the frustum edge
            y0, y1 = np.float32(0.7 * sgn), np.float32(1.5 * sgn)
            w = np.float32(1.0 + 0.01 * i)
            a = [xs[i] * w, y0 * w, 0.3 * w, w]; b = [xs[i + 1] * w, y0 * w, 0.3 * w, w]
            c = [xs[i] * w, y1 * w, 0.3 * w, w]; d = [xs[i + 1] * w, y1 * w, 0.3 * w, w]
            tris += [[a, b, c], [b, d, c]]
    tris = np.asarray(tris, np.float32)
    f = oracle.Frame(W, H)
    f.clear()
    setup = f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), keep_setup=True)
    assert setup.shape[0] > len(tris)                             # fans were emitted
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    g = r.setup_triangles()
    assert g.shape[0] == setup.shape[0]
    np.testing.assert_array_equal(g["spi"], setup["spi"])
    np.testing.assert_array_equal(g["rhw"].view(np.uint32), setup["rhw"].view(np.uint32))
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert_depth_equal(d, f.depth)


@pytest.mark.parametrize("blocked", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_owned_rows_are_what_a_partitioned_draw_writes(oracle, world, blocked):
    """frr_owned_rows (the slab a non-Python host gathers) == the rows a partitioned frr_draw defines: stitching
    the owned bands of every rank gives the unpartitioned image, and the bands tile the window exactly once."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 160, 203                                   # 7 tile rows, the last one 11 pixels high
    tris = scenes.random_clip_triangles(6000, W, H, seed=77, spread=1.1)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    got_d = np.full(W * H, -1.0, np.float32)
    got_t = np.zeros(W * H, np.uint32)
    seen = np.zeros(H, np.int32)
    r = fr.Renderer(W, H)
    m = r.upload_mesh(tris, fr.VS_CLIP)
    for rank in range(world):
        r.set_partition(rank, world, blocked=blocked)
        bands = r.owned_rows()
        assert all(a % 32 == 0 and a < b <= H for a, b in bands) and (len(bands) <= 1 or not blocked)
        r.clear()
        r.draw(m, fr.PS_DEPTH)
        _, d, t = r.readback()
        for a, b in bands:
            seen[a:b] += 1
            got_d[a * W:b * W] = d[a * W:b * W]
            got_t[a * W:b * W] = t[a * W:b * W]
    assert (seen == 1).all()
    np.testing.assert_array_equal(got_t, f.tri_id)
    np.testing.assert_array_equal(got_d.view(np.uint32), f.depth.view(np.uint32))
    assert r.owned_rows((0, 0)) == []


NAN_TRIS = np.array([
    [(-0.9, -0.9, 0.5, 1), (0.9, -0.9, 0.5, 1), (0.9, 0.9, 0.5, 1)],
    [(-0.9, -0.9, 0.5, 1), (0.9, 0.9, 0.5, 1), (-0.9, 0.9, 0.5, 1)],
    [(3e38, 0.2, 0.5, 1), (-0.5, -0.5, 0.5, 1), (-0.5, 0.6, 0.5, 1)],    # clipped; a vertex's screen x overflows to inf: every fragment of its fan is 0 * inf
    [(-0.3, -0.3, 1.0, 2), (0.7, -0.3, 1.0, 2), (0.2, 0.8, 1.0, 2)],      # farther (rhw 0.5), submitted AFTER the NaN fan: passes on the NaN pixels all the same
    [(-0.6, -0.1, 0.5, 4), (0.1, -0.1, 0.5, 4), (-0.2, 0.5, 0.5, 4)],     # farther still (rhw 0.25): must lose against the previous one again
], np.float32)


@pytest.mark.parametrize("path", ["span", "sweep", "span16"])
def test_nan_depth_fragments_follow_the_sequential_rule(oracle, monkeypatch, path):
    """renderer.rs:363-366: `if rhw < depth { continue }` is false when either side is NaN, so a fragment whose 1/w
    interpolates to NaN always passes, and the next fragment on that pixel passes too whatever its depth; after that the
    ordinary test resumes.  The tile kernels resolve depth without order, so a tile that saw a NaN fragment runs a second
    pass: per pixel only what was submitted after its last NaN fragment counts (tile_nan_begin).  Ids and depths must equal
    the sequential oracle's, NaN pixels included; then a second draw on top of the NaN depths left in the buffer, and a
    clean frame."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    if path == "sweep":
        monkeypatch.setenv("FRR_RASTER", "sweep")
    if path == "span16":
        monkeypatch.setenv("FRR_RASTER_NW", "16")
    W, H = 160, 120
    order = [0, 1, 2, 3, 4]
    for trial in range(3):
        tris = NAN_TRIS[order]
        f = oracle.Frame(W, H)
        f.clear()
        f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
        assert f.counters.frag_nan > 0 and (order[-1] != 2 or np.isnan(f.depth).any())
        r = fr.Renderer(W, H)
        r.clear()
        r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        assert r.last_warning is None
        st = r.stats()
        assert st["frag_nan"] == f.counters.frag_nan and st["tris_setup"] == f.counters.tris_setup and st["frag_covered"] == f.counters.frag_covered
        np.testing.assert_array_equal(t, f.tri_id)
        assert_depth_equal(d, f.depth)
        # a second draw over whatever the first left (NaN depths when the NaN fan came last): no clear in between
        more = scenes.random_clip_triangles(300, W, H, seed=17 + trial)
        f.draw(more, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms(), tri_id_base=int(f.counters.tris_setup))
        r.draw(r.upload_mesh(more, fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        assert_depth_equal(d, f.depth)
        r.close()
        order = [[3, 4, 0, 1, 2], [2, 0, 3, 1, 4]][trial % 2]   # NaN fan last (it owns its pixels, depth NaN); NaN fan first


def test_nan_depth_fragments_shaded_and_many_tiles(oracle):
    """The same rule through a shaded draw (the resolve re-evaluates the owner: NaN varyings quantise to 0 as Rust's
    `as u8` does) and with NaN fans sprinkled over a frame of many tiles, records enough per tile for the near-first
    copy (the second pass walks it) as well as for the register path."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 320, 224
    rng = np.random.default_rng(8)
    base = scenes.random_clip_triangles(40000, W, H, seed=23)
    bad = np.repeat(NAN_TRIS[2:3], 40, axis=0).copy()
    bad[:, 1:, 0] += rng.uniform(-0.4, 1.2, (40, 1)).astype(np.float32)
    bad[:, 1:, 1] += rng.uniform(-0.4, 0.4, (40, 1)).astype(np.float32)
    tris = np.concatenate([base[:20000], bad[:20], base[20000:], bad[20:]])
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(tris, oracle.VS_CLIP, oracle.PS_DEPTH, oracle.make_uniforms())
    assert f.counters.frag_nan > 1000
    r = fr.Renderer(W, H)
    r.clear()
    r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert_depth_equal(d, f.depth)
    assert r.stats()["frag_nan"] == f.counters.frag_nan
    # coloured vertices (VS_CLIP_COLOR: 7 floats per vertex): RGBA8 too
    col = np.concatenate([tris[:3000], tris[20000:20020]])
    colv = np.concatenate([col, rng.uniform(0, 1, col.shape[:2] + (3,)).astype(np.float32)], axis=2)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(colv, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
    assert f.counters.frag_nan > 0
    r.clear()
    r.draw(r.upload_mesh(colv, fr.VS_CLIP_COLOR), fr.PS_COLOR)
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    assert_depth_equal(d, f.depth)
    np.testing.assert_array_equal(c, f.color)


def test_set_option_rejects_unknown_names_and_values(oracle):
    """frr_set_option (include/frr.h): the library's development switches; it reads no environment variable itself (the
    Python binding translates FRR_* for the tests).  Unknown names and impossible values are FRR_ERR_INVALID, and a ctx
    keeps working afterwards."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 96, 64
    r = fr.Renderer(W, H)
    for name, value in (("no_such_switch", 1), ("raster_nw", 5), ("raster_occ", 7), ("bin_chunks", -1)):
        with pytest.raises(fr.FrrError) as e:
            r.set_option(name, value)
        assert e.value.code == fr.FRR_ERR_INVALID
    tris = scenes.random_clip_triangles(2000, W, H, seed=3)
    f = _oracle_depth(oracle, tris, W, H)
    for name, value in ((None, 0), ("raster_nw", 8), ("raster_sweep", 1), ("bin_atomics", 1), ("clear_eager", 1)):
        if name:
            r.set_option(name, value)
        r.clear()
        r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
        _, d, t = r.readback()
        np.testing.assert_array_equal(t, f.tri_id)
        assert_depth_equal(d, f.depth)
