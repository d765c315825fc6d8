"""Dev tool (GPU box, one rank): host time per frame of bench.py's distributed loop -- how long the Python side needs to
enqueue one partitioned frame + its gather, against the GPU time of a rank's frame (a host-bound loop caps what
bench.py --gpus N can show).  python tools/dist_host_cost.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
import bench
sys.argv = ["bench.py", "--force-dist", "--also", "none", "--cpu-baseline-seconds", "0", "--workload", sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
args = bench.parse_args()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
run = bench.Run(args, torch, dist, 0, 1, 0, bench.ALIASES.get(args.workload, args.workload))
with torch.cuda.stream(run.stream):
    for _ in range(10):
        run.step()
    run.drain(); torch.cuda.synchronize()
    for label, kw in (("render only", dict(gather=False)), ("render + gather (overlapped)", dict())):
        n = 300
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            run.step(**kw)
        t_host = time.perf_counter() - t0           # everything enqueued
        run.drain(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print(f"{label:32s} host enqueue {t_host / n * 1e6:7.1f} us/frame   until the GPU is done {t_all / n * 1e6:7.1f} us/frame", flush=True)
run.close()
dist.destroy_process_group()
