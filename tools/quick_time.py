"""Ad-hoc first timing of the headline workload with per-kernel HIP-event times (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f_renderer_amd as fr
from f_renderer_amd import scenes

W, H, n = 1920, 1080, 1_000_000
if len(sys.argv) > 1: W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tris = scenes.random_clip_triangles(n, W, H)
r = fr.Renderer(W, H)
m = r.upload_mesh(tris, fr.VS_CLIP)
if os.environ.get('PART'):
    rk, wd = map(int, os.environ['PART'].split(','))
    r.set_partition(rk, wd)
for _ in range(3):
    r.clear(); r.draw(m, fr.PS_DEPTH)
r.sync()
st_count = r.stats()
r.set_count_fragments(os.environ.get('COUNT','0')=='1')
K = 10
r.event_record(0)
for _ in range(K):
    r.clear(); r.draw(m, fr.PS_DEPTH)
r.event_record(1)
ms = r.event_elapsed_ms(0, 1) / K
st = st_count
print(f"{W}x{H} n={n}: frame {ms:.3f} ms  -> {n/ms/1e3:.1f} Mtri/s, {st['frag_covered']/ms/1e3:.1f} Mfrag/s", st)
r.profile_enable(True); r.profile_reset()
for _ in range(K):
    r.clear(); r.draw(m, fr.PS_DEPTH)
for k in fr.Renderer.KERNELS:
    t, c = r.profile_get(k)
    print(f"  {k:18s} {t/max(c,1)*1e3:9.1f} us  x{c}")
