"""Dev tool: survivor counts of the span kernel's early-z stages (needs a -DFRR_DEBUG_COUNTERS build:
FRR_LIB=tools/libfrr_dbg.so FRR_DEBUG_PRINT=1 python tools/debug_counters.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes
W, H, n = 1920, 1080, 1_000_000
if len(sys.argv) > 3: W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tris = scenes.random_clip_triangles(n, W, H)
r = fr.Renderer(W, H)
m = r.upload_mesh(tris, fr.VS_CLIP)
for count in (True, False):
    r.set_count_fragments(count)
    r.clear(); r.draw(m, fr.PS_DEPTH)
    print("count", count, "-> tri alive rows spans spans_live frags fwin rwin | zub-skippable frag-wins hiz-rebuilds", file=sys.stderr)
    r.stats()
