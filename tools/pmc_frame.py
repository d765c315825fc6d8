"""Dev tool: a few frames of one BASELINE config for rocprofv3 --pmc runs (no timing, no oracle).
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -d gpurun_out/pmcX -- python3 tools/pmc_frame.py [headline|cfg4|cfg5|...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes
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
for _ in range(int(os.environ.get("FRAMES", "4"))):
    r.clear(); r.draw(m, ps)
r.sync()
print(name, r.stats())
