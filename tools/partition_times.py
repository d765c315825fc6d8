"""Dev tool (one GPU): per-rank frame time of the tile partition for EVERY rank r of N (max and mean over the ranks,
render only, blocked layout, bound targets with two frames in flight as bench.py's multi-GPU loop uses them; SERIAL=1: one
frame at a time on the ctx's stream), N = 1, 2, 4, 8, on the headline, 4096^2 and 4K-sheets workloads.  Host wall clock.
  python tools/partition_times.py [--json profiles/r02_partition_times.json] [headline cfg4 cfg5]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402 (before the library: see tests/conftest.py)
import f_renderer_amd as fr
from f_renderer_amd import scenes

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json in args:
    args.remove(out_json)
rows = []
ST = torch.cuda.Stream()
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
    # as bench.py's multi-GPU loop renders: caller-bound target sets (three, rotating), two frames in flight on them
    # (option bound_targets_in_flight), the frame's exchange fenced on the caller's stream (no exchange here: one GPU)
    st = ST   # (one caller stream for every workload: a process's HIP streams share few hardware queues)
    sets = [(torch.zeros((H, W), dtype=torch.int32, device="cuda"), torch.zeros((H, W), dtype=torch.float32, device="cuda"),
             torch.zeros((H, W), dtype=torch.int32, device="cuda")) for _ in range(3)]
    r.set_option("bound_targets_in_flight", 0 if os.environ.get("SERIAL") else 1)

    def frame(i):
        c_, d_, t_ = sets[i % 3]
        r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
        r.clear(); r.draw(m, ps)
        r.frame_fence(st.cuda_stream)

    for N in (1, 2, 4, 8):
        per_rank = []
        for rank in range(N):
            r.set_partition(rank, N, blocked=True)
            for i in range(4):
                frame(i)
            r.sync()
            K = 40
            t0 = time.perf_counter()
            for i in range(K):
                frame(i)
            r.sync()
            per_rank.append((time.perf_counter() - t0) / K * 1e6)
            if os.environ.get("VERBOSE"):
                print("   rank", rank, "of", N, round(per_rank[-1], 1), "us, replays", r.stats()["replays"], flush=True)
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
