"""Dev tool (one GPU): per-rank frame time of the tile partition for EVERY rank r of N (max and mean over the ranks,
render only, blocked layout as bench.py uses it), N = 1, 2, 4, 8, on the headline, 4096^2 and 4K-sheets workloads.
  python tools/partition_times.py [--json profiles/r02_partition_times.json] [headline cfg4 cfg5]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
from f_renderer_amd import scenes

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json in args:
    args.remove(out_json)
rows = []
for name in (args or ["headline", "cfg4", "cfg5"]):
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
    for N in (1, 2, 4, 8):
        per_rank = []
        for rank in range(N):
            r.set_partition(rank, N, blocked=True)
            for _ in range(3):
                r.clear(); r.draw(m, ps)
            r.sync()
            K = 20
            r.event_record(0)
            for _ in range(K):
                r.clear(); r.draw(m, ps)
            r.event_record(1)
            per_rank.append(r.event_elapsed_ms(0, 1) / K * 1e3)
        row = dict(workload=name, width=W, height=H, N=N, max_us=round(max(per_rank), 1), mean_us=round(sum(per_rank) / N, 1),
                   per_rank_us=[round(x, 1) for x in per_rank])
        rows.append(row)
        print(json.dumps(row), flush=True)
    r.close()
for name in sorted(set(x["workload"] for x in rows)):
    one = [x for x in rows if x["workload"] == name]
    t1 = one[0]["max_us"]
    print(name, "render-only speed-up (1-GPU time / slowest rank):", {x["N"]: round(t1 / x["max_us"], 2) for x in one})
if out_json:
    json.dump(rows, open(out_json, "w"), indent=1)
