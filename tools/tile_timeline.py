"""Dev tool (GPU box, -DFRR_DEBUG_COUNTERS build): timeline of the tile kernel's workgroups on one frame -- when each
started and ended (100 MHz s_memrealtime), on which CU, how many ran at once, and where the wave cycles went.
  FRR_LIB=tools/libfrr_dbg.so FRR_DEBUG_TILES=1 FRR_DEBUG_PRINT=1 python tools/tile_timeline.py [headline|cfg4|cfg5]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f_renderer_amd as fr
from f_renderer_amd import scenes
from f_renderer_amd import _native as N

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
cfg = scenes.build_config(name)
W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
vs, ps = getattr(fr, "VS_" + cfg["vs"]), getattr(fr, "PS_" + cfg["ps"])
r = fr.Renderer(W, H)
if cfg["cam"]:
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
if cfg["tex"] is not None:
    r.set_texture(0, cfg["tex"]); r.set_uniforms(texture_slot=0)
m = r.upload_mesh(mesh, vs)
r.set_count_fragments(False)
for _ in range(4):
    r.clear(); r.draw(m, ps)
r.sync()
L = N.lib()
L.frr_debug_tiles.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
nt = C.c_uint32()
buf = np.zeros((((W + 31) // 32) * ((H + 31) // 32), 8), np.uint64)
rc = L.frr_debug_tiles(r._ctx, buf.ctypes.data, C.byref(nt))
assert rc == 0, rc
tl = buf[buf[:, 0] != 0]
t0 = tl[:, 0].min()
us = lambda x: (x.astype(np.int64) - int(t0)) / 100.0
start, main0, main1, end = us(tl[:, 0]), us(tl[:, 1]), us(tl[:, 2]), us(tl[:, 3])
hw = (tl[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (tl[:, 4] >> np.uint64(32)).astype(np.int64) & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
nent = (tl[:, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
print(f"{name}: {len(tl)} workgroups with work; kernel span {end.max():.1f} us; distinct CUs {len(set(cu.tolist()))}")
q = lambda a: " ".join(f"{np.percentile(a, p):7.1f}" for p in (0, 10, 50, 90, 99, 100))
print("percentiles           min     p10     p50     p90     p99     max")
print("start            ", q(start))
print("duration         ", q(end - start))
print("  pre-pass       ", q(main0 - start))
print("  main loop      ", q(main1 - main0))
print("  barrier+resolve", q(end - main1))
print("records per tile ", q(nent))
print("corr(duration, records) = %.3f" % np.corrcoef(end - start, nent)[0, 1])
edges = np.arange(0, end.max() + 5, 5.0)
act = [(int(((start <= t) & (end > t)).sum())) for t in edges]
print("active workgroups every 5 us:", act)
# concurrency per CU
mx = []
for c in set(cu.tolist()):
    s_, e_ = start[cu == c], end[cu == c]
    ev = sorted([(x, 1) for x in s_] + [(x, -1) for x in e_], key=lambda z: (z[0], z[1]))
    k = best = 0
    for _, d in ev:
        k += d; best = max(best, k)
    mx.append(best)
print("max concurrent workgroups per CU: min %d  median %d  max %d; workgroups per CU: min %d max %d" % (
    min(mx), int(np.median(mx)), max(mx), min(np.bincount(np.unique(cu, return_inverse=True)[1])), max(np.bincount(np.unique(cu, return_inverse=True)[1]))))
late = start > 5.0
print(f"workgroups starting after 5 us: {int(late.sum())}; their mean duration {np.mean((end - start)[late]) if late.any() else 0:.1f} vs early {np.mean((end - start)[~late]):.1f} us")
r.stats()
