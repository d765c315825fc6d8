"""Dev tool: one draw of a parity-test scene (ragged frame, clipped triangles), compared with the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f_renderer_amd as fr
from f_renderer_amd import scenes
from oracle import cref
W, H, n, spread, wj, seed = 333, 211, 20000, 1.15, 0.1, 2
if len(sys.argv) > 1:
    W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tris = scenes.random_clip_triangles(n, W, H, seed=seed, spread=spread, w_jitter=wj)
r = fr.Renderer(W, H)
r.set_count_fragments(os.environ.get("COUNT", "1") == "1")
r.clear()
r.draw(r.upload_mesh(tris, fr.VS_CLIP), fr.PS_DEPTH)
r.sync()
print("sync ok", flush=True)
_, d, t = r.readback()
f = cref.Frame(W, H); f.clear(); f.draw(tris, cref.VS_CLIP, cref.PS_DEPTH, cref.make_uniforms())
print("ids equal", np.array_equal(t, f.tri_id), "depth equal", np.array_equal(d.view(np.uint32), f.depth.view(np.uint32)), flush=True)
print("stats", r.stats(), f.counters.as_dict())
bad = np.nonzero(t != f.tri_id)[0]
print("bad ids", bad.size, [(int(i), int(t[i]), int(f.tri_id[i])) for i in bad[:12]])
bd = np.nonzero(d.view(np.uint32) != f.depth.view(np.uint32))[0]
print("bad depth", bd.size, [(int(i), float(d[i]), float(f.depth[i])) for i in bd[:8]])
