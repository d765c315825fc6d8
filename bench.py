#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rasterization path.

Metric (BASELINE.json): Mtri/s + Mfrag/s on a synthetic 1M-triangle frame @1920x1080.
A "step" = one frame of the hot path over resident inputs: frr_clear + frr_draw (geometry ->
binning -> tile raster/resolve) [+ the RCCL image gather when N > 1].  Inputs (the triangle
list) are in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: one process per GPU; the framebuffer is partitioned into N contiguous blocks of 32-px tile
rows (frr_set_partition_layout: blocked), geometry is replicated, and the only collective is one RCCL
gather of the owned slabs to rank 0 per frame, read straight from the render target and overlapped with
the next frame.  Total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (width, height, ntris, vs, ps)
    "random_1M_tris_1920x1080_depth": (1920, 1080, 1_000_000, "clip", "depth"),
    "random_1M_tris_4096x4096_depth": (4096, 4096, 1_000_000, "clip", "depth"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="random_1M_tris_1920x1080_depth", choices=sorted(WORKLOADS))
    ap.add_argument("--force-dist", action="store_true",
                    help="dev: initialise RCCL and run the band gather even with one rank (exercises the N>1 code path)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: wait for each frame's gather before rendering the next one (default: the gather of "
                         "frame i overlaps the rendering of frame i+1 into a second target set)")
    ap.add_argument("--gather-rows-div", type=int, default=1,
                    help="dev, with --force-dist on one rank: gather only 1/D of the rows (a band of the size a rank of D would send)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                    help="approximate CPU time budget of the oracle baseline leg (rank 0, N=1 only); 0 disables")
    return ap.parse_args()


def cpu_baseline(tris, W, H, budget_s):
    """Times the CPU oracle (oracle/frr_oracle.c, single thread) on whole frames of the SAME
    workload.  Returns (dict for the JSON line, oracle counters of one frame)."""
    from oracle import cref
    u = cref.make_uniforms()
    f = cref.Frame(W, H)
    reps, total, counters = 0, 0.0, None
    while reps == 0 or (total < budget_s and reps < 64):
        f.counters = cref.Counters()
        f.clear((30, 30, 30, 255), 0.0)
        t0 = time.perf_counter()
        f.clear((30, 30, 30, 255), 0.0)
        f.draw(tris, cref.VS_CLIP, cref.PS_DEPTH, u)
        total += time.perf_counter() - t0
        reps += 1
        counters = f.counters.as_dict()
    ntris = tris.shape[0]
    return {
        "value": ntris * reps / total / 1e6, "unit": "Mtri/s", "cores": 1, "kind": "port",
        "mfrag_per_s": counters["frag_covered"] * reps / total / 1e6,
        "sample": f"{reps} whole frame(s) of the same workload ({ntris} tris), clear+geometry+raster, "
                  f"{total:.1f} s on 1 of {os.cpu_count()} host cores; C restatement of renderer.rs "
                  f"(oracle/frr_oracle.c), optimistic: omits the reference's per-triangle heap allocations",
    }, counters


def load_pmc_traffic(workload):
    """HBM bytes per k_raster launch from a committed rocprofv3 --pmc run (profiles/*.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as fh:
            d = json.load(fh)
        e = d.get(workload, {}).get("k_raster")
        return float(e["hbm_bytes_per_launch"]) if e else None
    except Exception:
        return None


def main():
    args = parse_args()
    import torch
    import f_renderer_amd as fr
    from f_renderer_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dist = None
    saved_stdout_fd = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        # RCCL prints a version banner on file descriptor 1 when its communicator is created; this program's
        # stdout is ONE JSON line, so everything else written to fd 1 until then goes to stderr
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H, ntris, _, _ = WORKLOADS[args.workload]
    tris = scenes.random_clip_triangles(ntris, W, H)  # same arrays on every rank (deterministic)

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        r = fr.Renderer(W, H, device=local_rank, stream=stream.cuda_stream)
        # frame targets live in torch tensors (plumbing: device memory + the gather's operands);
        # rows are padded to a whole number of tile rows per rank so owned bands are one strided view
        from f_renderer_amd.multigpu import BlockGather, band_layout
        _, _, HP = band_layout(H, world)
        # three target sets: frame i renders into set i % 3 while the gathers of frames i-1 and i-2 may still be in flight;
        # before a set is reused the HOST checks that its gather (three frames back) has completed
        nsets = 3 if dist is not None else 1
        color = [torch.zeros((HP, W), dtype=torch.int32, device="cuda") for _ in range(nsets)]
        depth = [torch.zeros((HP, W), dtype=torch.float32, device="cuda") for _ in range(nsets)]
        tri_id = [torch.full((HP, W), -1, dtype=torch.int32, device="cuda") for _ in range(nsets)]
        r.bind_targets(color[0].data_ptr(), depth[0].data_ptr(), tri_id[0].data_ptr())
        r.set_partition(rank, world, blocked=True)   # contiguous slabs: the gather needs no staging copies
        dev_in = torch.from_numpy(tris).to("cuda")  # resident in HBM before timing
        mesh = r.bind_mesh_device(dev_in.data_ptr(), ntris, fr.VS_CLIP, keepalive=dev_in)
        gh = H // max(1, args.gather_rows_div) if world == 1 else H
        gathers = [BlockGather(gh, W, torch.float32, "cuda", rank, world) for _ in range(nsets)] if dist is not None else None
        ghp = band_layout(gh, world)[2]
        inflight = [None] * nsets
        final = None
        frame_no = 0

        def drain():
            nonlocal final
            for s in range(nsets):
                if inflight[s] is not None:
                    final = gathers[s].finish(inflight[s])
                    inflight[s] = None

        def step():
            nonlocal final, frame_no
            s = frame_no % nsets
            frame_no += 1
            if gathers is not None:
                if inflight[s] is not None:            # the gather that last read this target set must be done
                    final = gathers[s].finish(inflight[s]) if args.sync_gather else gathers[s].finish_host(inflight[s])
                r.bind_targets(color[s].data_ptr(), depth[s].data_ptr(), tri_id[s].data_ptr())
            r.clear((30, 30, 30, 255), 0.0)
            r.draw(mesh, fr.PS_DEPTH)
            if gathers is not None:
                # owned bands -> ONE RCCL gather to rank 0 (final image only), overlapped with the next frame
                inflight[s] = gathers[s].start(depth[s][:ghp])
                if args.sync_gather:
                    final = gathers[s].finish(inflight[s])
                    inflight[s] = None

        # one counted frame: the exact covered-fragment count of this rank's tiles (the Mfrag/s
        # numerator).  The statistic is then switched off: maintaining it forbids the tile kernel's
        # whole-triangle early-z (outputs are identical either way, tests/test_gpu_earlyz.py).
        r.set_count_fragments(True)
        step()
        drain()
        r.sync()
        counted = r.stats()
        r.set_count_fragments(False)
        for _ in range(args.warmup):
            step()
        drain()
        r.sync()
        stats = r.stats()
        if stats["overflow"] or counted["overflow"]:
            raise SystemExit("device work list overflow during warmup")

        r.profile_reset()
        PROF_PERIOD = 4  # HIP events around the dominant kernel only, every 4th launch (a pair costs the stream ~4 us)
        r.profile_enable(True, kernels=["k_raster"], period=PROF_PERIOD)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()                      # every frame's image has reached rank 0 inside the timed region
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        raster_ms, raster_n = r.profile_get("k_raster")
        r.profile_enable(False)
        stats = r.stats()

        # multi-GPU image check on rank 0: gathered image == what a single full render would hold
        image_ok = None
        if gathers is not None and rank == 0:
            # every pixel of the gathered depth image must come from its owner's render (clear value 0 or a
            # positive 1/w): finite and non-negative everywhere, and not all background
            img = final[:gh]
            image_ok = bool(torch.isfinite(img).all().item() and (img >= 0).all().item() and (img > 0).any().item())

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    cov = torch.tensor([float(counted["frag_covered"])], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cov, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    frag_covered = int(cov.item())  # covered fragments of one frame over all ranks' tiles

    if saved_stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        os.close(saved_stdout_fd)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mtri = ntris / (ms_per_step * 1e-3) / 1e6
        mfrag = frag_covered / (ms_per_step * 1e-3) / 1e6
        line = {
            "metric": "Mtri/s (and Mfrag/s), 1M-triangle frame @1920x1080" if H == 1080 else "Mtri/s (and Mfrag/s), 1M-triangle frame",
            "value": round(mtri, 3), "unit": "Mtri/s", "mfrag_per_s": round(mfrag, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": W, "height": H, "triangles": ntris,
                       "setup_triangles": stats["tris_setup"], "covered_fragments": frag_covered,
                       "shader": "VS_CLIP/PS_DEPTH (depth-only)", "tile": "32x32",
                       "partition": f"tile rows in {world} contiguous block(s)" + (", RCCL gather to rank 0" if world > 1 else "")},
        }
        if image_ok is not None:
            line["gathered_image_finite"] = image_ok
        cpu, oc = None, None
        if world == 1 and args.cpu_baseline_seconds > 0:
            cpu, oc = cpu_baseline(tris, W, H, args.cpu_baseline_seconds)
            line["cpu_baseline"] = cpu
            line["parity_counts_match"] = bool(oc["frag_covered"] == frag_covered and oc["tris_setup"] == stats["tris_setup"])
            line["speedup_vs_cpu_1core"] = round(mtri / cpu["value"], 1)
        # roofline of the dominant kernel (k_raster): algorithmic bytes per launch:
        #   N_setup * (108 + 12K) record bytes read by the raster pass          (K = 0 here)
        # + 4 B depth read per covered fragment + 8 B (depth+colour/id write) per z-passing fragment
        if raster_n:
            avg_ms = raster_ms / raster_n
            f_pass = oc["frag_zpass"] if oc else None
            if world == 1 and f_pass is not None:
                alg = stats["tris_setup"] * 108 + 4 * frag_covered + 8 * f_pass
                ach = alg / (avg_ms * 1e-3) / 1e9
                line["roofline"] = {"bound": "hbm", "kernel": "k_raster", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                    "traffic": load_pmc_traffic(args.workload),
                                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(avg_ms, 5),
                                    "launches": raster_n, "sampled_every": PROF_PERIOD, "frag_zpass": f_pass}
            else:
                line["roofline"] = {"bound": "hbm", "kernel": "k_raster", "achieved": None, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                                    "launches": raster_n,
                                    "note": "per-rank launch covers 1/N of the tiles; algorithmic bytes are quoted at N=1"}
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
