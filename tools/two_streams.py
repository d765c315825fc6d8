"""Dev experiment: two contexts on two HIP streams rendering alternate frames -- does the GPU overlap the
HBM-bound geometry of one frame with the VALU-bound tile kernel of the other?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import f_renderer_amd as fr
from f_renderer_amd import scenes
W, H, n = 1920, 1080, 1_000_000
tris = scenes.random_clip_triangles(n, W, H)
dev_in = torch.from_numpy(tris).to("cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
rs = []
for s in streams:
    r = fr.Renderer(W, H, device=0, stream=s.cuda_stream)
    r.set_count_fragments(False)
    rs.append((r, r.bind_mesh_device(dev_in.data_ptr(), n, fr.VS_CLIP, keepalive=dev_in)))
def run(k, nctx):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        r, m = rs[i % nctx]
        r.clear(); r.draw(m, fr.PS_DEPTH)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
for nctx in (1, 2):
    run(20, nctx)
    print(f"{nctx} context(s): {run(200, nctx):.1f} us/frame")
rs[0][0].profile_enable(True, kernels=["k_raster"])
run(20, 1)
print(f"1 context, HIP events around the tile kernel: {run(200, 1):.1f} us/frame")
