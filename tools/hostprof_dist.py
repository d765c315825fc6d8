import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
import f_renderer_amd as fr
from f_renderer_amd import scenes
from f_renderer_amd.multigpu import BandGather, band_layout
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H, n = 1920, 1080, 1_000_000
tris = scenes.random_clip_triangles(n, W, H)
stream = torch.cuda.Stream()
T = {}
def tick(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
with torch.cuda.stream(stream):
    r = fr.Renderer(W, H, device=0, stream=stream.cuda_stream)
    _, _, HP = band_layout(H, 1)
    color = [torch.zeros((HP, W), dtype=torch.int32, device="cuda") for _ in range(2)]
    depth = [torch.zeros((HP, W), dtype=torch.float32, device="cuda") for _ in range(2)]
    tid = [torch.full((HP, W), -1, dtype=torch.int32, device="cuda") for _ in range(2)]
    dev_in = torch.from_numpy(tris).to("cuda")
    mesh = r.bind_mesh_device(dev_in.data_ptr(), n, fr.VS_CLIP, keepalive=dev_in)
    gh = H // 8
    g = [BandGather(gh, W, torch.float32, "cuda", 0, 1) for _ in range(2)]
    ghp = band_layout(gh, 1)[2]
    r.set_count_fragments(False)
    infl = [None, None]
    for it in range(220):
        if it == 20:
            torch.cuda.synchronize(); T.clear(); t_all = time.perf_counter()
        s = it % 2
        t0 = time.perf_counter()
        if infl[s] is not None: g[s].finish(infl[s])
        tick("finish", t0); t0 = time.perf_counter()
        r.bind_targets(color[s].data_ptr(), depth[s].data_ptr(), tid[s].data_ptr())
        tick("bind", t0); t0 = time.perf_counter()
        r.clear((30, 30, 30, 255), 0.0)
        tick("clear", t0); t0 = time.perf_counter()
        r.draw(mesh, fr.PS_DEPTH)
        tick("draw", t0); t0 = time.perf_counter()
        v = depth[s][:ghp]
        tick("slice", t0); t0 = time.perf_counter()
        infl[s] = g[s].start(v)
        tick("start", t0)
    t_issue = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t_all
print("per-step host issue %.1f us, total %.1f us" % (t_issue / 200 * 1e6, t_total / 200 * 1e6))
for k, v in T.items(): print("  %-8s %.1f us" % (k, v / 200 * 1e6))
dist.destroy_process_group()
