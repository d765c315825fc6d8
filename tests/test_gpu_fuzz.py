"""Seeded fuzz of the HIP path against the oracle: random frame sizes, triangle soups (clipped fans, w = 0 vertices, NaN-depth
fragments), library options, partitions and layouts.  Nobody re-issues a frame here: a work list that is too small is the
library's business (FRR_ERR_CAPACITY never reaches the caller).  tools/fuzz_vs_oracle.py runs the same cases in bulk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_case(fr, scenes, cref, rng, small_lists=False):
    """One random scene on 1..5 ranks; returns (ok, description, replays, had NaN fragments)."""
    from f_renderer_amd.multigpu import tile_row_owner
    W = int(rng.integers(1, 700)); H = int(rng.integers(1, 500))
    n = int(rng.integers(1, 30000))
    spread = float(rng.uniform(0.8, 2.5)); wj = float(rng.choice([0.1, 0.5, 1.2, 2.0]))
    seed = int(rng.integers(0, 1 << 30))
    tris = scenes.random_clip_triangles(n, W, H, seed=seed, spread=spread, w_jitter=wj)
    if rng.random() < 0.2:
        tris[:: max(1, n // 7), int(rng.integers(0, 3)), 3] = 0.0
    if rng.random() < 0.2:                                   # vertices whose screen position overflows: fans of NaN-depth fragments
        tris[:: max(1, n // 5), int(rng.integers(0, 3)), 0] = 3e38
    opts = {"clip_queue": int(rng.integers(-1, 2)), "raster_nw": int(rng.choice([0, 0, 3, 4, 8, 16])), "overlap": int(rng.integers(0, 3)), "frames_in_flight": int(rng.integers(1, 3))}
    if small_lists:                                          # force replays: tiny work lists
        opts["bin_capacity"] = int(rng.integers(64, 4000))
        opts["fan_capacity"] = int(rng.integers(8, 512))
    f = cref.Frame(W, H); f.clear((3, 2, 1, 0), 0.0)
    f.draw(tris, cref.VS_CLIP, cref.PS_DEPTH, cref.make_uniforms())
    has_nan = bool(f.counters.frag_nan)   # (NaN depth fragments follow the reference's sequential rule: compared, NaN == NaN)
    G = int(rng.integers(1, 6)); blocked = bool(rng.integers(0, 2))
    owner = np.asarray(tile_row_owner((H + 31) // 32, G, blocked))[np.arange(H) // 32]
    acc_t = np.full(W * H, 0xFFFFFFFF, np.uint32); acc_d = np.zeros(W * H, np.float32); cov = 0
    desc = f"W={W} H={H} n={n} spread={spread:.3f} wj={wj} seed={seed} G={G} blocked={blocked} opts={opts}"
    ok, replays = True, 0
    for rank in range(G):
        r = fr.Renderer(W, H)
        for k_, v_ in opts.items():
            r.set_option(k_, v_)
        if G > 1:
            r.set_partition(rank, G, blocked=blocked)
        r.set_count_fragments(True)
        m = r.upload_mesh(tris, fr.VS_CLIP)
        r.clear((3, 2, 1, 0), 0.0); r.draw(m, fr.PS_DEPTH)
        _, d, t = r.readback()
        st = r.stats(); r.close()
        replays += st["replays"]
        own = np.repeat(owner == rank, W)
        acc_t[own] = t[own]; acc_d[own] = d[own]; cov += st["frag_covered"]
        ok = ok and st["tris_setup"] == f.counters.tris_setup and st["tris_in"] == n
    gn, wn = np.isnan(acc_d), np.isnan(f.depth)
    ok = ok and np.array_equal(acc_t, f.tri_id) and np.array_equal(gn, wn) and \
        np.array_equal(acc_d.view(np.uint32)[~gn], f.depth.view(np.uint32)[~wn]) and cov == f.counters.frag_covered
    return ok, desc, replays, has_nan


@pytest.mark.parametrize("small_lists", [False, True])
def test_fuzz_against_the_oracle(oracle, small_lists):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    rng = np.random.default_rng(31337 if small_lists else 2024)
    bad, replays = [], 0
    for _ in range(24 if small_lists else 40):
        ok, desc, rp, _ = run_case(fr, scenes, oracle, rng, small_lists)
        replays += rp
        if not ok:
            bad.append(desc)
    assert not bad, bad
    assert replays > 0 or not small_lists     # the tiny lists really were too small somewhere
