"""Dev tool (GPU box): seeded fuzz of SHADED frames against the oracle -- vertex colours through the clipper (K = 3 varyings),
two draws per frame (two meshes: triangle ids continue across draws), random frame sizes, options and partitions, with and
without tiny work lists (FUZZ_SMALL=1: the draws are verified and replayed inside the library).  RGBA8, depth bits and ids
must equal the oracle's.  FUZZ_N cases, FUZZ_SEED."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes
from f_renderer_amd.multigpu import tile_row_owner
from oracle import cref

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "77")))
N = int(os.environ.get("FUZZ_N", "100"))
small = bool(int(os.environ.get("FUZZ_SMALL", "0")))
bad = replays = 0
for it in range(N):
    W = int(rng.integers(8, 600)); H = int(rng.integers(8, 400))
    meshes = []
    for k in range(2):
        n = int(rng.integers(1, 12000))
        clip = scenes.random_clip_triangles(n, W, H, seed=int(rng.integers(0, 1 << 30)), spread=float(rng.uniform(0.8, 2.0)), w_jitter=float(rng.choice([0.1, 0.5, 1.2])))
        col = scenes.splitmix_u01(int(rng.integers(0, 1 << 30)), n * 9).reshape(n, 3, 3).astype(np.float32)
        meshes.append(np.concatenate([clip, col], axis=2))
    f = cref.Frame(W, H); f.clear((5, 6, 7, 8), 0.0)
    base = 0
    for m in meshes:
        f.draw(m, cref.VS_CLIP_COLOR, cref.PS_COLOR, cref.make_uniforms(), tri_id_base=base)
        base = int(f.counters.tris_setup)
    if f.counters.frag_nan:
        continue          # (NaN depths: a NaN colour's bit pattern is the machine's; the depth-only fuzz covers those frames)
    opts = {"clip_queue": int(rng.integers(-1, 2)), "raster_nw": int(rng.choice([0, 0, 4, 8, 16])), "overlap": int(rng.integers(0, 3)),
            "frames_in_flight": int(rng.integers(1, 3))}
    if small:
        opts["bin_capacity"] = int(rng.integers(64, 4000)); opts["fan_capacity"] = int(rng.integers(8, 512))
    G = int(rng.integers(1, 4)); blocked = bool(rng.integers(0, 2))
    owner = np.asarray(tile_row_owner((H + 31) // 32, G, blocked))[np.arange(H) // 32]
    acc_c = np.zeros((H, W, 4), np.uint8); acc_t = np.zeros(W * H, np.uint32); acc_d = np.zeros(W * H, np.float32)
    for rank in range(G):
        r = fr.Renderer(W, H)
        for k_, v_ in opts.items():
            r.set_option(k_, v_)
        if G > 1:
            r.set_partition(rank, G, blocked=blocked)
        ms = [r.upload_mesh(m, fr.VS_CLIP_COLOR) for m in meshes]
        for frame in range(2):            # the second frame repeats the first: proven passes, no verification wait
            r.clear((5, 6, 7, 8), 0.0)
            for m in ms:
                r.draw(m, fr.PS_COLOR)
        c, d, t = r.readback()
        replays += r.stats()["replays"]; r.close()
        own = owner == rank
        acc_c[own] = c[own]; acc_t.reshape(H, W)[own] = t.reshape(H, W)[own]; acc_d.reshape(H, W)[own] = d.reshape(H, W)[own]
    ok = np.array_equal(acc_t, f.tri_id) and np.array_equal(acc_d.view(np.uint32), f.depth.view(np.uint32)) and np.array_equal(acc_c, f.color)
    if not ok:
        bad += 1; print("MISMATCH", it, W, H, [m.shape[0] for m in meshes], opts, G, blocked, flush=True)
    if it % 10 == 0:
        print("case", it, "bad so far", bad, flush=True)
print("shaded fuzz done:", N, "cases,", bad, "bad, replays seen at the end of the second frame:", replays)
