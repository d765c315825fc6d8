"""Dev tool (GPU box): frame time and per-kernel time (HIP events) of the BASELINE configs with the library FRR_LIB points
at (default: the in-tree build), plus an image hash -- for A/B runs of build variants.
  [FRR_LIB=/path/lib.so] [PART=rank,world] python tools/time_configs.py [headline cfg4 cfg5 cfg3 ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes

tag = os.environ.get("TAG", os.path.basename(os.environ.get("FRR_LIB", "default")))
for name in (sys.argv[1:] or ["headline", "cfg4", "cfg5"]):
    cfg = scenes.build_config(name)
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    vs, ps = getattr(fr, "VS_" + cfg["vs"]), getattr(fr, "PS_" + cfg["ps"])
    r = fr.Renderer(W, H)
    if os.environ.get("PART"):                       # PART=rank,world: one rank of the blocked tile partition
        rank, world = (int(x) for x in os.environ["PART"].split(","))
        r.set_partition(rank, world, blocked=True)
    if cfg["cam"]:
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    if cfg["tex"] is not None:
        r.set_texture(0, cfg["tex"]); r.set_uniforms(texture_slot=0)
    r.set_uniforms(flat_color=cfg["flat_color"])
    m = r.upload_mesh(mesh, vs)
    r.set_count_fragments(False)
    for _ in range(3):
        r.clear(); r.draw(m, ps)
    r.sync()
    best = 1e9
    for rep in range(3):
        K = 20
        r.event_record(0)
        for _ in range(K):
            r.clear(); r.draw(m, ps)
        r.event_record(1)
        best = min(best, r.event_elapsed_ms(0, 1) / K)
    c, d, t = r.readback()
    h = hashlib.sha256(c.tobytes() + d.tobytes() + t.tobytes()).hexdigest()[:10]
    r.profile_enable(True); r.profile_reset()
    for _ in range(10):
        r.clear(); r.draw(m, ps)
    ks = []
    for k in fr.Renderer.KERNELS:
        tt, n = r.profile_get(k)
        if n:
            ks.append(f"{k[2:]} {tt / n * 1e3:.1f}")
    print(f"{tag:14s} {name:9s} frame {best * 1e3:7.1f} us | " + " | ".join(ks) + f" | {h}", flush=True)
    r.close()
