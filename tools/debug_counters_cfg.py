"""Dev tool: tile-kernel funnel counters for the camera scenes (needs the -DFRR_DEBUG_COUNTERS build):
FRR_LIB=tools/libfrr_dbg.so FRR_DEBUG_PRINT=1 python tools/debug_counters_cfg.py cfg3|cfg5|cfg2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
W, H, mesh, vs, ps = {"cfg2": (1920, 1080, scenes.torus, fr.VS_GOURAUD, fr.PS_COLOR),
                      "cfg3": (1920, 1080, scenes.displaced_sphere, fr.VS_PHONG, fr.PS_PHONG),
                      "cfg5": (3840, 2160, scenes.layered_sheets, fr.VS_PHONG, fr.PS_BLINN)}[which]
eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
r = fr.Renderer(W, H)
r.set_texture(0, scenes.checker_texture(1024, 32))
r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
m = r.upload_mesh(mesh(), vs)
r.set_count_fragments(False)
r.clear(); r.draw(m, ps)
print(which, "-> tri alive rows spans spans_live frags fwin rwin | zub-skippable frag-wins hiz-rebuilds", file=sys.stderr)
print(r.stats())
