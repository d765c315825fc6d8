"""Dev tool (GPU box, one rank): host time of one asynchronous torch.distributed collective call, the candidates for the
per-frame slab exchange (the loop of bench.py --gpus N is host-bound by it)."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29518")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
H, W = 1088, 1920
img = torch.zeros((H, W), dtype=torch.float32, device="cuda")
final = torch.zeros((H, W), dtype=torch.float32, device="cuda")
gathered = list(final.view(1, H, W).unbind(0))
side = torch.cuda.Stream()
def bench(label, fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hs = [fn() for _ in range(n)]
    t = time.perf_counter() - t0
    for h in hs:
        if h is not None: h.wait()
    torch.cuda.synchronize()
    print(f"{label:44s} {t / n * 1e6:7.1f} us per call (host)", flush=True)
bench("dist.gather(list, async)", lambda: dist.gather(img[0:H], gathered, dst=0, async_op=True))
bench("dist.gather(no slice, async)", lambda: dist.gather(img, gathered, dst=0, async_op=True))
bench("all_gather_into_tensor(async)", lambda: dist.all_gather_into_tensor(final, img, async_op=True))
bench("dist.broadcast(async)", lambda: dist.broadcast(img, 0, async_op=True))
bench("dist.all_reduce(async)", lambda: dist.all_reduce(img, async_op=True))
bench("isend+irecv self (batch)", lambda: dist.batch_isend_irecv([dist.P2POp(dist.isend, img, 0), dist.P2POp(dist.irecv, final, 0)])[0])
dist.destroy_process_group()
