// gather_rccl.cpp -- the final-image exchange of a multi-GPU run from a NON-Python host: one process per GPU, the
// framebuffer partitioned into contiguous blocks of tile rows through the reference's own sub-window seam
// (Renderer::rasterization's height_range, /root/reference/f_renderer/src/renderer.rs:270-271), and ONE RCCL group of
// ncclSend / ncclRecv per frame that moves every rank's slab straight from its render targets into the final image on
// rank 0 (xGMI; a slab is one contiguous range of each row-major plane: frr_owned_rows + frr_target_ptrs).
//
//   gather_rccl [--world N --rank R --id-file PATH] [--width W --height H --tris T] [--frames F]
//
// world 1 (the default) needs no id file.  For N > 1 start N processes (rank R uses HIP device R); rank 0 writes the
// ncclUniqueId to PATH, the others wait for it.  Rank 0 finally renders the same frame unpartitioned with a second
// context and compares: depth bits and triangle ids of the gathered image must equal frr_readback's, byte for byte.
// Exit code 0 = equal.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/frr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 2; } } while (0)
#define CHECK_FRR(c, x) do { int r_ = (x); if (r_ != FRR_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, frr_last_error(c)); return 2; } } while (0)

// the scenes' generator (SplitMix64): every rank builds the same triangle list
static uint64_t sm_state;
static float u01() { uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (float)(z >> 40) * (1.0f / 16777216.0f); }

static std::vector<float> random_clip_triangles(uint32_t n, uint32_t W)
{
    std::vector<float> v((size_t)n * 12);
    sm_state = 0xF5EED004ull;
    for (uint32_t t = 0; t < n; ++t) {
        const float wc = 1.0f + 9.0f * u01(), cx = 0.95f * (2.0f * u01() - 1.0f), cy = 0.95f * (2.0f * u01() - 1.0f);
        const float rpx = 2.0f * powf(16.0f, u01()), rn = 2.0f * rpx / (float)W;
        for (int k = 0; k < 3; ++k) {
            const float x = cx + rn * (2.0f * u01() - 1.0f), y = cy + rn * (2.0f * u01() - 1.0f), w = wc * (1.0f + 0.1f * (2.0f * u01() - 1.0f));
            float *o = &v[((size_t)t * 3 + k) * 4];
            o[0] = x * w; o[1] = y * w; o[2] = 0.5f * w; o[3] = w;
        }
    }
    return v;
}

int main(int argc, char **argv)
{
    int world = 1, rank = 0, frames = 3;
    uint32_t W = 640, H = 500, T = 20000;
    std::string id_file;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--world") world = atoi(argv[i + 1]);
        else if (k == "--rank") rank = atoi(argv[i + 1]);
        else if (k == "--id-file") id_file = argv[i + 1];
        else if (k == "--width") W = (uint32_t)atoi(argv[i + 1]);
        else if (k == "--height") H = (uint32_t)atoi(argv[i + 1]);
        else if (k == "--tris") T = (uint32_t)atoi(argv[i + 1]);
        else if (k == "--frames") frames = atoi(argv[i + 1]);
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && id_file.empty())) { fprintf(stderr, "bad --world/--rank/--id-file\n"); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 3; }
    CHECK_HIP(hipSetDevice(rank % ndev));

    // communicator: one rank per process (ncclCommInitRank), the id travels through a file
    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        if (!id_file.empty()) {
            FILE *f = fopen((id_file + ".tmp").c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", id_file.c_str()); return 2; }
            fclose(f);
            rename((id_file + ".tmp").c_str(), id_file.c_str());
        }
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "rank %d: no id file %s\n", rank, id_file.c_str()); return 2; }
        fclose(f);
    }
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // this rank's context: its block of tile rows, targets owned by the library
    frr_ctx *ctx = nullptr;
    if (frr_create(rank % ndev, W, H, stream, &ctx) != FRR_OK) { fprintf(stderr, "frr_create failed (needs a gfx950 device)\n"); return 3; }
    CHECK_FRR(ctx, frr_set_partition(ctx, rank, world));
    CHECK_FRR(ctx, frr_set_partition_layout(ctx, 1));
    const std::vector<float> tris = random_clip_triangles(T, W);
    int mesh = -1;
    CHECK_FRR(ctx, frr_mesh_upload(ctx, tris.data(), T, FRR_VS_CLIP, &mesh));

    // the final image lives on rank 0
    float *final_depth = nullptr; uint32_t *final_ids = nullptr;
    const size_t npx = (size_t)W * H;
    if (rank == 0) { CHECK_HIP(hipMalloc((void **)&final_depth, npx * 4)); CHECK_HIP(hipMalloc((void **)&final_ids, npx * 4)); }
    const uint8_t clear_rgba[4] = {30, 30, 30, 255};
    for (int frame = 0; frame < frames; ++frame) {
        CHECK_FRR(ctx, frr_clear(ctx, clear_rgba, 0.0f));                              // phong.rs:316-317
        CHECK_FRR(ctx, frr_draw(ctx, mesh, FRR_PS_DEPTH, 0, (int32_t)W, 0, (int32_t)H)); // loops A + B, phong.rs:321-381
        void *color, *depth, *ids;
        CHECK_FRR(ctx, frr_target_ptrs(ctx, &color, &depth, &ids));                    // device pointers, row stride W * 4 bytes
        (void)color;
        // ONE group per frame, on the ctx's stream (so it is ordered after the frame's tile kernels; and the library orders
        // the frame that next renders into this target set -- two frames on -- behind what this stream holds by then).
        // The operations are frr_exchange_plan's: a pure function of (window, partition, rank), tested for every rank of
        // worlds 2, 4 and 8 without that many GPUs (tests/test_multigpu_gloo.py)
        frr_xfer ops[64];
        const int nops = frr_exchange_plan(0, (int32_t)H, W, rank, world, /*blocked*/ 1, /*root*/ 0, ops, 64);
        if (nops < 0 || nops > 64) { fprintf(stderr, "frr_exchange_plan: %d\n", nops); return 2; }
        CHECK_NCCL(ncclGroupStart());
        for (int k = 0; k < nops; ++k) {
            const frr_xfer &x = ops[k];
            if (x.kind == FRR_XFER_SEND) {
                CHECK_NCCL(ncclSend((const float *)depth + x.offset, x.count, ncclFloat, x.peer, comm, stream));
                CHECK_NCCL(ncclSend((const uint32_t *)ids + x.offset, x.count, ncclUint32, x.peer, comm, stream));
            } else if (x.kind == FRR_XFER_RECV) {
                CHECK_NCCL(ncclRecv(final_depth + x.offset, x.count, ncclFloat, x.peer, comm, stream));
                CHECK_NCCL(ncclRecv(final_ids + x.offset, x.count, ncclUint32, x.peer, comm, stream));
            }
        }
        CHECK_NCCL(ncclGroupEnd());
        for (int k = 0; k < nops; ++k) {   // the root's own slab: a device copy
            const frr_xfer &x = ops[k];
            if (x.kind != FRR_XFER_COPY) continue;
            CHECK_HIP(hipMemcpyAsync(final_depth + x.offset, (const float *)depth + x.offset, x.count * 4, hipMemcpyDeviceToDevice, stream));
            CHECK_HIP(hipMemcpyAsync(final_ids + x.offset, (const uint32_t *)ids + x.offset, x.count * 4, hipMemcpyDeviceToDevice, stream));
        }
    }
    CHECK_FRR(ctx, frr_sync(ctx));
    CHECK_HIP(hipStreamSynchronize(stream));

    int rc = 0;
    if (rank == 0) {
        // the same frame, unpartitioned, through frr_readback
        frr_ctx *ref = nullptr;
        if (frr_create(0, W, H, nullptr, &ref) != FRR_OK) return 3;
        int m2 = -1;
        CHECK_FRR(ref, frr_mesh_upload(ref, tris.data(), T, FRR_VS_CLIP, &m2));
        CHECK_FRR(ref, frr_clear(ref, clear_rgba, 0.0f));
        CHECK_FRR(ref, frr_draw(ref, m2, FRR_PS_DEPTH, 0, (int32_t)W, 0, (int32_t)H));
        std::vector<float> want_d(npx), got_d(npx);
        std::vector<uint32_t> want_t(npx), got_t(npx);
        CHECK_FRR(ref, frr_readback(ref, nullptr, want_d.data(), want_t.data()));
        CHECK_HIP(hipMemcpy(got_d.data(), final_depth, npx * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(got_t.data(), final_ids, npx * 4, hipMemcpyDeviceToHost));
        const bool eq = memcmp(want_d.data(), got_d.data(), npx * 4) == 0 && memcmp(want_t.data(), got_t.data(), npx * 4) == 0;
        size_t drawn = 0;
        for (uint32_t t : got_t) drawn += t != 0xFFFFFFFFu;
        printf("gather_rccl: world %d, %ux%u, %u triangles, %d frame(s): gathered image %s the unpartitioned render (%zu pixels drawn)\n",
               world, W, H, T, frames, eq ? "EQUALS" : "DIFFERS FROM", drawn);
        rc = eq && drawn ? 0 : 1;
        frr_destroy(ref);
        (void)hipFree(final_depth); (void)hipFree(final_ids);
    }
    frr_destroy(ctx);
    ncclCommDestroy(comm);
    (void)hipStreamDestroy(stream);
    return rc;
}
