"""Dev tool: host enqueue cost and frame time of the headline frame with the second stream on and off (option overlap)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes

cfg = scenes.build_config(os.environ.get("CFG", "headline"))
W, H = cfg["W"], cfg["H"]
N = int(os.environ.get("N", "200"))
modes = [(int(os.environ["OVERLAP"]), int(os.environ.get("FIF", "2")))] if os.environ.get("OVERLAP") else [(0, 1), (2, 1), (1, 1), (0, 2)]
for overlap, fif in modes:
    r = fr.Renderer(W, H)
    r.set_option("overlap", overlap)
    r.set_option("frames_in_flight", fif)
    if os.environ.get("PART"):
        rk, wd = (int(x) for x in os.environ["PART"].split(","))
        r.set_partition(rk, wd, blocked=True)
    if cfg["tex"] is not None:
        r.set_texture(0, cfg["tex"])
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    r.set_count_fragments(False)
    m = r.upload_mesh(cfg["mesh"], getattr(fr, "VS_" + cfg["vs"]))
    ps = getattr(fr, "PS_" + cfg["ps"])
    bound = int(os.environ.get("BOUND", "0"))    # BOUND=1: caller-bound targets (three sets, rotating, as bench.py's multi-GPU loop binds them); 2: + option bound_targets_in_flight
    if bound:
        st = torch.cuda.Stream()
        sets = [(torch.zeros((H, W), dtype=torch.int32, device="cuda"), torch.zeros((H, W), dtype=torch.float32, device="cuda"),
                 torch.zeros((H, W), dtype=torch.int32, device="cuda")) for _ in range(3)]
        if bound == 2:
            r.set_option("bound_targets_in_flight", 1)

    def frame(i):
        if bound:
            c_, d_, t_ = sets[i % 3]
            r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
        r.clear(); r.draw(m, ps)
        if bound == 2:
            r.frame_fence(st.cuda_stream)       # (where the exchange of the frame would be enqueued)

    for i in range(10):
        frame(i)
    r.sync()
    t0 = time.perf_counter()
    for i in range(N):
        frame(i)
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"{os.environ.get('CFG', 'headline')} bound={os.environ.get('BOUND', '0')} part={os.environ.get('PART', '-')} overlap={overlap} frames_in_flight={fif}: host enqueue {(t1 - t0) / N * 1e6:.1f} us/frame, frame {(t2 - t0) / N * 1e6:.1f} us, replays {r.stats()['replays']}", flush=True)
    r.close()
