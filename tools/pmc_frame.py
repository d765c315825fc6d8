"""Dev tool: a few headline frames for rocprofv3 --pmc runs (no timing, no oracle).
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -d gpurun_out/pmcX -- python3 tools/pmc_frame.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes
W, H, n = 1920, 1080, 1_000_000
tris = scenes.random_clip_triangles(n, W, H)
r = fr.Renderer(W, H)
m = r.upload_mesh(tris, fr.VS_CLIP)
r.set_count_fragments(False)
for _ in range(int(os.environ.get("FRAMES", "4"))):
    r.clear(); r.draw(m, fr.PS_DEPTH)
r.sync()
print(r.stats())
