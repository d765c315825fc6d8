#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rasterization path.

Metric (BASELINE.json): Mtri/s + Mfrag/s on a synthetic 1M-triangle frame @1920x1080.
A "step" = one frame of the hot path over resident inputs: frr_clear + frr_draw (geometry ->
binning -> tile raster/resolve) [+ the RCCL image gather when N > 1].  Inputs (the triangle
list, textures) are in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W [--workload headline|cfg2|cfg3|cfg4|cfg5] [--also cfg4,cfg5]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: one process per GPU; the framebuffer is partitioned into N contiguous blocks of 32-px tile
rows (frr_set_partition_layout: blocked), geometry is replicated, and the only collective is the RCCL
gather of the owned slabs (depth for depth-only workloads, colour + depth for shaded ones) to rank 0 per
frame, read straight from the render targets and overlapped with the next frames.  Before the timed region
rank 0 renders the same frame unpartitioned; the gathered image of the LAST timed frame must equal it byte
for byte (depth bits, RGBA8, and -- in one extra check frame -- triangle ids).  Total work is fixed =>
"scaling": "strong".

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).  `--also` (default: the 4096^2
and the 4K textured configs north_star's scaling target is quoted on) runs further workloads after the primary
one, fewer steps each, and reports them under "secondary" with the same fields.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# name -> (config of f_renderer_amd.scenes.build_config, text)
WORKLOADS = {
    "random_1M_tris_1920x1080_depth": ("headline", "1M random clip-space triangles, 1920x1080, depth-only"),
    "random_1M_tris_4096x4096_depth": ("cfg4", "1M random clip-space triangles, 4096x4096, depth-only"),
    "sheets_259k_tris_3840x2160_blinn": ("cfg5", "250,000-triangle layered sheets, 3840x2160, textured Blinn-Phong"),
    "sphere_69k_tris_1920x1080_phong": ("cfg3", "69,192-triangle displaced sphere, 1920x1080, textured Phong"),
    "torus_6k_tris_1920x1080_gouraud": ("cfg2", "6,272-triangle torus, 1920x1080, Gouraud"),
}
ALIASES = {cfg: name for name, (cfg, _) in WORKLOADS.items()}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROF_PERIOD = 4             # timed region: k_raster launches are bracketed by HIP events only every 4th time (an event pair costs the stream ~4 us)
ISO_LAUNCHES = 16           # launches of the dominant kernel timed one frame at a time (the primary roofline figure)
SIMDS, CLOCK_GHZ = 1024, 2.4   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2400 MHz max clock (the VALU-issue bound is priced at the max clock)
DEFAULT_FIF = int(os.environ.get("FRR_FRAMES_IN_FLIGHT", "2"))   # what f_renderer_amd.Renderer sets from the environment (dev runs)
DEFAULT_OVERLAP = int(os.environ.get("FRR_OVERLAP", "2"))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="random_1M_tris_1920x1080_depth", choices=sorted(WORKLOADS) + sorted(ALIASES))
    ap.add_argument("--also", default="cfg4,cfg5",
                    help="comma-separated further workloads reported under 'secondary' (fewer steps each); 'none' disables")
    ap.add_argument("--in-flight", action="store_true",
                    help="N = 1 only: also time two contexts on two streams rendering alternate frames (extra key "
                         "two_frames_in_flight; off by default so that a kernel trace of the default command holds "
                         "one-frame-at-a-time launches only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="dev: initialise RCCL and run the slab gather even with one rank (exercises the N>1 code path)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: wait for each frame's gather before rendering the next one (default: the gathers of "
                         "frames i-1, i-2 overlap the rendering of frame i into other target sets)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                    help="approximate CPU time budget of the oracle baseline leg (rank 0, N=1 only); 0 disables")
    return ap.parse_args()


def source_sha():
    """Digest of the kernel sources the loaded library was built from (profiles/pmc_traffic.json entries carry the
    digest they were measured on; a stale entry is not reported)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "f_renderer_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc_traffic(workload):
    """(HBM bytes per k_raster launch, VALU-issue entry or None, stale?) from a committed rocprofv3 --pmc run
    (profiles/pmc_traffic.json, written by tools/make_pmc_traffic.py; reported only while the kernel sources are the ones
    the counters were measured on)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            d = json.load(fh)
        e = d.get(workload, {})
        k = e.get("k_raster")
        if not k:
            return None, None, False
        if e.get("_source_sha") != source_sha():
            return None, None, True
        return float(k["hbm_bytes_per_launch"]), k.get("valu_issue"), False
    except Exception:
        return None, None, False


def golden_counts(cfg_name):
    """tris_setup / frag_covered / frag_zpass of the full-size config from the committed digests (data only)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as fh:
            return json.load(fh)[cfg_name]["full"]
    except Exception:
        return None


def cpu_baseline(cfg, budget_s):
    """Times the CPU oracle (oracle/frr_oracle.c, single thread) on whole frames of the SAME
    workload.  Returns (dict for the JSON line, oracle counters of one frame, all-core best-effort dict)."""
    from oracle import cref
    from f_renderer_amd import scenes
    W, H, mesh = cfg["W"], cfg["H"], cfg["mesh"]
    vs, ps = getattr(cref, "VS_" + cfg["vs"]), getattr(cref, "PS_" + cfg["ps"])
    kw = {}
    if cfg["cam"]:
        eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
        kw = dict(view=cref.set_look_at(eye, at, up), proj=cref.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    if cfg["tex"] is not None:
        kw["tex"] = cref.Texture(cfg["tex"])
    u = cref.make_uniforms(flat_color=cfg["flat_color"], **kw)
    f = cref.Frame(W, H)
    reps, total, counters = 0, 0.0, None
    while reps == 0 or (total < budget_s and reps < 64):
        f.counters = cref.Counters()
        f.clear((30, 30, 30, 255), 0.0)
        t0 = time.perf_counter()
        f.clear((30, 30, 30, 255), 0.0)
        f.draw(mesh, vs, ps, u)
        total += time.perf_counter() - t0
        reps += 1
        counters = f.counters.as_dict()
    ntris = mesh.shape[0]
    base = {
        "value": ntris * reps / total / 1e6, "unit": "Mtri/s", "cores": 1, "kind": "port",
        "mfrag_per_s": counters["frag_covered"] * reps / total / 1e6,
        "sample": f"{reps} whole frame(s) of the same workload ({ntris} tris), clear+geometry+raster, "
                  f"{total:.1f} s on 1 of {os.cpu_count()} host cores; C restatement of renderer.rs "
                  f"(oracle/frr_oracle.c), optimistic: omits the reference's per-triangle heap allocations",
    }
    # SURVEY 8d's "best-effort all-core CPU": the same code with the rows split over host threads through the
    # reference's own sub-window argument; NOT the baseline (the reference is single-threaded)
    threads = min(os.cpu_count() or 1, 64)
    _, _, _, cov, secs = cref.draw_banded(W, H, mesh, vs, ps, u, threads=threads)
    best = {"value": ntris / secs / 1e6, "unit": "Mtri/s", "cores": threads, "seconds": round(secs, 3),
            "note": "row bands over host threads via the sub-window argument (renderer.rs:270-271); every band repeats "
                    "the geometry of all triangles, which is what the reference's per-triangle API offers; not the baseline"}
    return base, counters, best


def two_frames_in_flight(torch, run, steps):
    """N = 1 extra: the same frames rendered by TWO contexts on two HIP streams, alternately (application-level frames
    in flight, as any real-time renderer keeps them): the HBM-bound geometry and binning of frame i+1 fill the drain of
    frame i's tile kernel.  Reported beside `value` (which stays the one-frame-at-a-time figure the roofline is
    measured on), never instead of it."""
    fr = run.fr
    ctxs = []
    for _ in range(2):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            r = fr.Renderer(run.W, run.H, device=run.local_rank, stream=st.cuda_stream)
            cfg = run.cfg
            from f_renderer_amd import scenes
            kw = {}
            if cfg["cam"]:
                eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(run.W, run.H)
                kw = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
            if cfg["tex"] is not None:
                r.set_texture(0, cfg["tex"])
                kw["texture_slot"] = 0
            r.set_uniforms(flat_color=cfg["flat_color"], **kw)
            r.set_count_fragments(False)
            m = r.bind_mesh_device(run.dev_in.data_ptr(), run.ntris, run.vs, keepalive=run.dev_in)
            ctxs.append((r, m))

    def loop(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            r, m = ctxs[i & 1]
            r.clear((30, 30, 30, 255), 0.0)
            r.draw(m, run.ps)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    loop(6)
    el = loop(steps)
    for r, _ in ctxs:
        r.close()
    ms = el / steps * 1e3
    return {"frames_in_flight": 2, "contexts": 2, "ms_per_step": round(ms, 5), "value": round(run.ntris / (ms * 1e-3) / 1e6, 3),
            "unit": "Mtri/s", "steps": steps}


class Run:
    """One workload on this rank: contexts, targets, gathers."""

    def __init__(self, args, torch, dist, rank, world, local_rank, name):
        import f_renderer_amd as fr
        from f_renderer_amd import scenes
        from f_renderer_amd.multigpu import FrameGather, band_layout
        self.args, self.torch, self.dist, self.rank, self.world, self.fr = args, torch, dist, rank, world, fr
        self.name = name
        self.cfg_name = WORKLOADS[name][0]
        cfg = self.cfg = scenes.build_config(self.cfg_name)        # same arrays on every rank (deterministic)
        W, H = self.W, self.H = cfg["W"], cfg["H"]
        self.ntris = int(cfg["mesh"].shape[0])
        self.vs, self.ps = getattr(fr, "VS_" + cfg["vs"]), getattr(fr, "PS_" + cfg["ps"])
        self.K = fr.lib().frr_vs_num_varyings(self.vs)
        self.shaded = cfg["ps"] != "DEPTH"
        # ONE torch stream for every workload of this process: a process's HIP streams share a handful of hardware queues, and
        # the library's second frame stream must not end up on the same queue as the caller's
        if not hasattr(Run, "_stream"):
            Run._stream = torch.cuda.Stream()
        self.stream = Run._stream
        with torch.cuda.stream(self.stream):
            # frame targets live in torch tensors (plumbing: device memory + the gather's operands); the height is
            # padded to a whole number of tile rows per rank so that every rank's slab has the same size
            _, _, HP = band_layout(H, world)
            self.HP = HP
            # three target sets: frame i renders into set i % 3 while the gathers of frames i-1 and i-2 may still be
            # in flight; before a set is reused the HOST checks that its gather (three frames back) has completed
            self.nsets = 3 if dist is not None else 1
            mk = lambda dt, fill: [torch.full((HP, W), fill, dtype=dt, device="cuda") for _ in range(self.nsets)]  # noqa: E731
            self.color, self.depth, self.tri_id = mk(torch.int32, 0), mk(torch.float32, 0.0), mk(torch.int32, -1)
            self.r = self.make_renderer(local_rank)
            if dist is not None:   # (N = 1: the library's own targets -- two sets, so that two frames are in flight)
                self.r.bind_targets(self.color[0].data_ptr(), self.depth[0].data_ptr(), self.tri_id[0].data_ptr())
                # N > 1: the same two frames in flight on the caller-bound target sets (three of them, rotating): frames run on
                # the library's private streams, the exchange of a frame is fenced on this stream (frr_frame_fence)
                self.r.set_option("bound_targets_in_flight", 1)
            self.r.set_partition(rank, world, blocked=True)   # contiguous slabs: the gather needs no staging copies
            self.dev_in = torch.from_numpy(cfg["mesh"]).to("cuda")  # resident in HBM before timing
            self.mesh = self.r.bind_mesh_device(self.dev_in.data_ptr(), self.ntris, self.vs, keepalive=self.dev_in)
            self.gathers, self.gather_ids = None, None
            if dist is not None:
                planes = ([(torch.int32, ())] if self.shaded else []) + [(torch.float32, ())]
                self.gathers = [FrameGather(H, W, planes, "cuda", rank, world) for _ in range(self.nsets)]
                self.gather_ids = FrameGather(H, W, [(torch.int32, ())], "cuda", rank, world)
            self.inflight = [None] * self.nsets
            self.final = None
            self.frame_no = 0
            self.local_rank = local_rank

    def make_renderer(self, local_rank):
        fr, cfg = self.fr, self.cfg
        from f_renderer_amd import scenes
        r = fr.Renderer(self.W, self.H, device=local_rank, stream=self.stream.cuda_stream)
        kw = {}
        if cfg["cam"]:
            eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(self.W, self.H)
            kw = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
        if cfg["tex"] is not None:
            r.set_texture(0, cfg["tex"])
            kw["texture_slot"] = 0
        r.set_uniforms(flat_color=cfg["flat_color"], **kw)
        return r

    def planes_of(self, s):
        return ([self.color[s]] if self.shaded else []) + [self.depth[s]]

    def drain(self):
        for s in range(self.nsets):
            if self.inflight[s] is not None:
                self.final = self.gathers[s].finish(self.inflight[s])
                self.inflight[s] = None

    def step(self, gather=True, sync_gather=None):
        sync_gather = self.args.sync_gather if sync_gather is None else sync_gather
        s = self.frame_no % self.nsets
        self.frame_no += 1
        g = self.gathers if gather else None
        if self.gathers is not None:
            if self.inflight[s] is not None:            # the gather that last read this target set must be done
                if sync_gather:
                    # stream-side completion only: this stream now waits for the exchange, and the frame about to render into
                    # the set runs on one of the library's private streams -- its first write of the targets waits for this stream
                    self.final = self.gathers[s].finish(self.inflight[s])
                    self.r.frame_wait(self.stream.cuda_stream)
                else:
                    self.final = self.gathers[s].finish_host(self.inflight[s])   # (the host has seen it complete)
                self.inflight[s] = None
            self.r.bind_targets(self.color[s].data_ptr(), self.depth[s].data_ptr(), self.tri_id[s].data_ptr())
        self.r.clear((30, 30, 30, 255), 0.0)
        self.r.draw(self.mesh, self.ps)
        if g is not None:
            # owned slabs -> the RCCL exchange to rank 0 (final image only), overlapped with the next frames: this stream
            # waits for the frame (no host wait), the exchange follows it
            self.r.frame_fence(self.stream.cuda_stream)
            self.inflight[s] = g[s].start(self.planes_of(s))
            if sync_gather:
                self.final = g[s].finish(self.inflight[s])
                self.inflight[s] = None
        return s

    def reference_image(self):
        """Rank 0: the same frame rendered unpartitioned by a second context -> (colour, depth, ids) tensors."""
        torch = self.torch
        r = self.make_renderer(self.local_rank)
        c = torch.zeros((self.H, self.W), dtype=torch.int32, device="cuda")
        d = torch.zeros((self.H, self.W), dtype=torch.float32, device="cuda")
        t = torch.full((self.H, self.W), -1, dtype=torch.int32, device="cuda")
        r.bind_targets(c.data_ptr(), d.data_ptr(), t.data_ptr())
        m = r.bind_mesh_device(self.dev_in.data_ptr(), self.ntris, self.vs, keepalive=self.dev_in)
        r.set_count_fragments(False)
        r.clear((30, 30, 30, 255), 0.0)
        r.draw(m, self.ps)
        r.sync()
        r.close()
        return c, d, t

    def timed_loop(self, n, **kw):
        torch, dist = self.torch, self.dist
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            self.step(**kw)
        self.drain()                      # every frame's image has reached rank 0 inside the timed region
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def run(self, steps, warmup, primary):
        torch, dist, r = self.torch, self.dist, self.r
        with torch.cuda.stream(self.stream):
            ref = self.reference_image() if (self.gathers is not None and self.rank == 0) else None
            # one counted frame: the exact covered-fragment count of this rank's tiles (the Mfrag/s
            # numerator).  The statistic is then switched off: maintaining it forbids the tile kernel's
            # whole-triangle early-z (outputs are identical either way, tests/test_gpu_earlyz.py).
            r.set_count_fragments(True)
            self.step()
            self.drain()
            r.sync()
            counted = r.stats()
            r.set_count_fragments(False)
            for _ in range(warmup):
                self.step()
            self.drain()
            r.sync()
            stats = r.stats()

            # One frame at a time (what a caller that reads every frame back gets: the reference's loop is serial,
            # phong.rs:314-387): the frame period, and the dominant kernel's launches with nothing beside them -- the
            # primary roofline figure.  (In the timed region below the library keeps two frames in flight: there the
            # kernel shares the chip with the next frame's geometry + binning, and consecutive tile kernels may overlap.)
            self._raster_iso, self._serial_ms = (0.0, 0), None
            if self.gathers is None:
                r.set_option("frames_in_flight", 1)
                r.set_option("overlap", 0)
                for _ in range(3):
                    self.step()
                r.sync()
                n1 = max(8, min(steps, 20))
                self._serial_ms = self.timed_loop(n1) / n1 * 1e3
                r.profile_reset()
                r.profile_enable(True, kernels=["k_raster"], period=1)
                for _ in range(ISO_LAUNCHES):
                    self.step()
                    r.sync()
                self._raster_iso = r.profile_get("k_raster")
                r.profile_enable(False)
                r.set_option("frames_in_flight", DEFAULT_FIF)
                r.set_option("overlap", DEFAULT_OVERLAP)
                for _ in range(3):
                    self.step()
                r.sync()

            r.profile_reset()
            r.profile_enable(True, kernels=["k_raster"], period=PROF_PERIOD)
            elapsed = self.timed_loop(steps)
            raster_ms, raster_n = r.profile_get("k_raster")
            r.profile_enable(False)
            stats = r.stats()

            # N = 1: the image the LAST timed frame left in the targets against the committed digests of the oracle's
            # frame (tests/golden/frames.json: data only, nothing of oracle/ runs here)
            self._golden_match = None
            gold = golden_counts(self.cfg_name)
            if self.world == 1 and gold:
                sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
                if self.gathers is None:
                    c_, d_, t_ = r.readback()
                else:
                    r.sync()
                    s_last, H = (self.frame_no - 1) % self.nsets, self.H
                    c_, d_, t_ = (x[s_last][:H].contiguous().cpu().numpy() for x in (self.color, self.depth, self.tri_id))
                self._golden_match = bool(sha(d_) == gold["sha256_depth"] and sha(t_) == gold["sha256_tri_id"] and sha(c_) == gold["sha256_rgba8"])

            # multi-GPU image check on rank 0: the gathered image of the LAST timed frame == the unpartitioned render,
            # byte for byte; triangle ids through one extra (untimed) frame's gather
            image_equal = None
            render_us = gather_us = None
            if self.gathers is not None:
                s = self.step(gather=False)
                self.r.frame_fence(self.stream.cuda_stream)
                ids_final = self.gather_ids([self.tri_id[s]])
                torch.cuda.synchronize()
                if self.rank == 0:
                    H = self.H
                    bits = lambda x: x.view(torch.int32)  # noqa: E731
                    got = list(self.final)
                    want = ([ref[0]] if self.shaded else []) + [ref[1]]
                    image_equal = all(bool(torch.equal(bits(g[:H]), bits(w))) for g, w in zip(got, want))
                    image_equal = image_equal and bool(torch.equal(ids_final[0][:H], ref[2]))
                # render and gather separately (max over ranks): a render-only loop, then the same loop with a gather
                # that is waited for every frame (serial, nothing overlapped)
                n2 = max(4, min(steps, 20))
                t_render = self.timed_loop(n2, gather=False)
                t_both = self.timed_loop(n2, sync_gather=True)
                render_us, gather_us = t_render / n2 * 1e6, max(0.0, (t_both - t_render) / n2 * 1e6)

        t = torch.tensor([elapsed, render_us or 0.0, gather_us or 0.0], dtype=torch.float64, device="cuda")
        cov = torch.tensor([float(counted["frag_covered"])], dtype=torch.float64, device="cuda")
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(cov, op=dist.ReduceOp.SUM)
        elapsed = float(t[0].item())
        frag_covered = int(cov.item())  # covered fragments of one frame over all ranks' tiles
        if self.rank != 0:
            return None
        world, W, H, ntris = self.world, self.W, self.H, self.ntris
        ms_per_step = elapsed / steps * 1e3
        mtri = ntris / (ms_per_step * 1e-3) / 1e6
        mfrag = frag_covered / (ms_per_step * 1e-3) / 1e6
        gathered = ("colour + depth" if self.shaded else "depth")
        out = {
            "value": round(mtri, 3), "unit": "Mtri/s", "mfrag_per_s": round(mfrag, 1),
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 5),
            "config": {"workload": self.name, "scene": WORKLOADS[self.name][1], "width": W, "height": H, "triangles": ntris,
                       "setup_triangles": stats["tris_setup"] if world == 1 else None, "covered_fragments": frag_covered,
                       "shader": f"VS_{self.cfg['vs']}/PS_{self.cfg['ps']}", "varyings": self.K, "tile": "32x32",
                       "frames_in_flight": DEFAULT_FIF,   # (N > 1: on the caller-bound target sets, option bound_targets_in_flight)
                       "partition": f"tile rows in {world} contiguous block(s)" + (f", RCCL gather of {gathered} to rank 0" if self.gathers is not None else "")},
        }
        if self._golden_match is not None:
            out["image_matches_golden"] = self._golden_match
        if image_equal is not None:
            out["gathered_image_equal"] = image_equal
            out["render_us_max"] = round(float(t[1].item()), 2)
            out["gather_us_max"] = round(float(t[2].item()), 2)
        self._raster = (raster_ms, raster_n)
        self._stats = stats
        self._frag_covered = frag_covered
        return out

    def roofline(self, out, f_pass, n_setup):
        """Roofline of the dominant kernel (k_raster).  Algorithmic bytes per launch:
          N_setup * (108 + 12K) record bytes read by the raster pass
        + 4 B depth read per covered fragment + 8 B (depth + colour/id write) per z-passing fragment;
        `frac_fragment_pass` is the literal fragment-pass form (4 F_cov + 8 F_pass only).  The primary figures come from
        launches with nothing beside them (one frame at a time); `in_flight` is the same kernel in the timed region."""
        raster_ms, raster_n = self._raster
        iso_ms, iso_n = self._raster_iso
        ms_per_step = out["ms_per_step"]
        if self._serial_ms is not None:
            out["ms_per_frame_serial"] = round(self._serial_ms, 5)
        if self.world == 1 and f_pass is not None and (iso_n or raster_n):
            K, F_cov = self.K, self._frag_covered
            nf = self.fr.lib().frr_vs_input_floats(self.vs)
            alg = n_setup * (108 + 12 * K) + 4 * F_cov + 8 * f_pass
            frag_alg = 4 * F_cov + 8 * f_pass
            # the whole frame's algorithmic bytes (B_alg) and what the frame period makes of them
            frame_alg = self.ntris * nf * 12 + 2 * n_setup * (108 + 12 * K) + frag_alg + 8 * self.W * self.H
            out["frame_alg_bytes"] = frame_alg
            out["frame_alg_gbs"] = round(frame_alg / (ms_per_step * 1e-3) / 1e9, 1)
            traffic, valu, stale = load_pmc_traffic(self.name)
            fig = lambda ms: {"avg_launch_ms": round(ms, 5), "achieved": round(alg / (ms * 1e-3) / 1e9, 1),  # noqa: E731
                              "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_fragment_pass": round(frag_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            prim_ms, prim_n, mode = (iso_ms / iso_n, iso_n, "one frame at a time") if iso_n else (raster_ms / raster_n, raster_n, "timed region")
            roof = {"bound": "hbm", "kernel": "k_raster", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg, "fragment_pass_bytes_per_launch": frag_alg, "frag_zpass": f_pass,
                    "launches": prim_n, "measured": mode}
            roof.update(fig(prim_ms))
            if raster_n and iso_n:
                avg = raster_ms / raster_n
                if avg <= ms_per_step:
                    roof["in_flight"] = dict(fig(avg), launches=raster_n, sampled_every=PROF_PERIOD)
                else:   # consecutive frames' tile kernels overlapped: the event bracket timed a time-shared kernel
                    roof["in_flight"] = {"dropped": "avg_launch_ms %.5f > ms_per_step: overlapping launches" % avg, "launches": raster_n}
            roof["note"] = ("achieved = ALGORITHMIC bytes (the reference's memory semantics) / launch time, an efficiency figure; "
                            "the kernel resolves depth in LDS, its measured HBM bytes are `traffic`; the kernel is bound by VALU "
                            "issue, see `valu_issue`")
            if valu:
                # VALU-issue bound: instructions the launch issues (PMC) x the measured issue cost per instruction kind
                # (profiles/*valu_issue_rates.txt, weighted by the static mix of the kernel's hot loops) / SIMD cycles available
                cyc = valu["insts_valu"] * valu["cycles_per_valu"]
                roof["valu_issue"] = {"bound": "valu_issue", "insts_valu": valu["insts_valu"], "insts_salu": valu.get("insts_salu"),
                                      "cycles_per_valu": valu["cycles_per_valu"], "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                                      "frac": round(cyc / (SIMDS * prim_ms * 1e-3 * CLOCK_GHZ * 1e9), 4)}
            if stale:
                roof["traffic_stale"] = "profiles/pmc_traffic.json was measured on other kernel sources"
            out["roofline"] = roof
        elif raster_n:
            out["roofline"] = {"bound": "hbm", "kernel": "k_raster", "achieved": None, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": round(raster_ms / raster_n, 5),
                               "launches": raster_n,
                               "note": "per-rank launch covers 1/N of the tiles; algorithmic bytes are quoted at N=1"}

    def close(self):
        self.r.close()


def main():
    args = parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_dist:
        # a rank keeps two frames in flight on the library's two private streams beside this program's own stream and RCCL's:
        # HIP maps a process's streams onto 4 hardware queues by default, and two streams that share one run nothing beside
        # each other (measured: a rank of 8 of the 4K textured frame 80 us with 4 queues, 64 us with 8).  Before HIP starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch

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

    primary = ALIASES.get(args.workload, args.workload)
    also = [] if args.also.strip().lower() in ("", "none") else [ALIASES.get(a.strip(), a.strip()) for a in args.also.split(",")]
    also = [a for a in also if a in WORKLOADS and a != primary]

    run = Run(args, torch, dist, rank, world, local_rank, primary)
    line = run.run(args.steps, args.warmup, True)
    if rank == 0:
        H = run.H
        head = {"metric": "Mtri/s (and Mfrag/s), 1M-triangle frame @1920x1080" if run.cfg_name == "headline" else "Mtri/s (and Mfrag/s) per frame",
                "value": line["value"], "unit": "Mtri/s", "mfrag_per_s": line["mfrag_per_s"], "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": line["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        head.update({k: v for k, v in line.items() if k not in head})
        line = head
        gold = golden_counts(run.cfg_name)
        f_pass, n_setup = (gold["frag_zpass"], gold["tris_setup"]) if gold else (None, None)
        if world == 1 and args.cpu_baseline_seconds > 0:
            cpu, oc, best = cpu_baseline(run.cfg, args.cpu_baseline_seconds)
            line["cpu_baseline"] = cpu
            line["cpu_best_effort"] = best
            line["parity_counts_match"] = bool(oc["frag_covered"] == run._frag_covered and oc["tris_setup"] == run._stats["tris_setup"]
                                               and (gold is None or (gold["frag_zpass"] == oc["frag_zpass"] and gold["frag_covered"] == oc["frag_covered"])))
            line["speedup_vs_cpu_1core"] = round(line["value"] / cpu["value"], 1)
            f_pass, n_setup = oc["frag_zpass"], oc["tris_setup"]
        run.roofline(line, f_pass, n_setup)
    if rank == 0 and world == 1 and dist is None and args.in_flight:
        extra = two_frames_in_flight(torch, run, args.steps)
        if extra:
            line["two_frames_in_flight"] = extra
    run.close()
    del run

    secondary = []
    for name in also:
        steps2 = max(4, min(args.steps, 50))
        r2 = Run(args, torch, dist, rank, world, local_rank, name)
        o = r2.run(steps2, min(args.warmup, 3), False)
        if rank == 0:
            gold = golden_counts(r2.cfg_name)
            if gold:
                r2.roofline(o, gold["frag_zpass"], gold["tris_setup"])
                o["parity_counts_match"] = bool(gold["frag_covered"] == r2._frag_covered)
            secondary.append(o)
        r2.close()
        del r2

    if saved_stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        os.close(saved_stdout_fd)
    if rank == 0:
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
