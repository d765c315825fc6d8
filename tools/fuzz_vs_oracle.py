"""Dev tool: random small frames (sizes, triangle soups, partitions, layouts) against the oracle; prints every mismatch."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes
from oracle import cref
rng = np.random.default_rng(2024)
bad = 0
N = int(os.environ.get("FUZZ_N", "150"))
for it in range(N):
    W = int(rng.integers(1, 700)); H = int(rng.integers(1, 500))
    n = int(rng.integers(1, 30000))
    spread = float(rng.uniform(0.8, 2.5)); wj = float(rng.choice([0.1, 0.5, 1.2, 2.0]))
    seed = int(rng.integers(0, 1 << 30))
    tris = scenes.random_clip_triangles(n, W, H, seed=seed, spread=spread, w_jitter=wj)
    if rng.random() < 0.2:
        tris[:: max(1, n // 7), int(rng.integers(0, 3)), 3] = 0.0
    if rng.random() < 0.2:                                   # vertices whose screen position overflows: fans of NaN-depth fragments
        tris[:: max(1, n // 5), int(rng.integers(0, 3)), 0] = 3e38
    opts = {"clip_queue": int(rng.integers(-1, 2)), "raster_nw": int(rng.choice([0, 0, 3, 4, 8, 16]))}
    f = cref.Frame(W, H); f.clear((3, 2, 1, 0), 0.0)
    f.draw(tris, cref.VS_CLIP, cref.PS_DEPTH, cref.make_uniforms())
    has_nan = bool(f.counters.frag_nan)   # (NaN depth fragments follow the reference's sequential rule: compared, NaN == NaN)
    G = int(rng.integers(1, 6)); blocked = bool(rng.integers(0, 2))
    rows = np.arange(H) // 32; k = -(-((H + 31) // 32) // G)
    acc_t = np.full(W * H, 0xFFFFFFFF, np.uint32); acc_d = np.zeros(W * H, np.float32); cov = 0
    for rank in range(G):
        r = fr.Renderer(W, H)
        for k_, v_ in opts.items(): r.set_option(k_, v_)
        if G > 1: r.set_partition(rank, G, blocked=blocked)
        r.set_count_fragments(bool(rng.integers(0, 2)) or True)
        m = r.upload_mesh(tris, fr.VS_CLIP)
        for attempt in range(4):
            r.clear((3, 2, 1, 0), 0.0); r.draw(m, fr.PS_DEPTH)
            try:
                _, d, t = r.readback()
                break
            except fr.FrrError as e:               # documented: a work list overflowed, capacity grown, re-issue the frame
                assert e.code == fr.FRR_ERR_CAPACITY, e
                print("capacity re-issue", it, attempt, flush=True)
        st = r.stats(); r.close()
        own = np.repeat(((rows // k) == rank) if (blocked and G > 1) else ((rows % G) == rank), W)
        acc_t[own] = t[own]; acc_d[own] = d[own]; cov += st["frag_covered"]
        if st["tris_setup"] != f.counters.tris_setup: bad += 1; print("SETUP MISMATCH", it, W, H, n, seed, G, blocked)
    gn, wn = np.isnan(acc_d), np.isnan(f.depth)
    ok = np.array_equal(acc_t, f.tri_id) and np.array_equal(gn, wn) and np.array_equal(acc_d.view(np.uint32)[~gn], f.depth.view(np.uint32)[~wn]) and cov == f.counters.frag_covered
    nan_cases = globals().get("nan_cases", 0) + int(has_nan)
    if it % 10 == 0:
        print("case", it, "bad so far", bad, flush=True)
    if not ok:
        bad += 1; print("MISMATCH", it, W, H, n, spread, wj, seed, G, blocked, opts, flush=True)
print("fuzz done:", N, "cases,", nan_cases, "with NaN fragments,", bad, "bad")
