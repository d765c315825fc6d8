"""Two frames in flight: consecutive frames run on two streams with their own target sets, workspace sets and device-table
lanes (own targets: default; caller-bound targets: option bound_targets_in_flight + frr_frame_fence).  Alternating two
DIFFERENT scenes from frame to frame, without a host synchronisation in between, every frame must come out as the oracle's."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes(oracle, W, H):
    from f_renderer_amd import scenes
    out = []
    for k, (n, spread) in enumerate(((9000, 1.1), (2500, 1.6))):
        clip = scenes.random_clip_triangles(n, W, H, seed=30 + k, spread=spread, w_jitter=0.4)
        col = scenes.splitmix_u01(90 + k, n * 9).reshape(n, 3, 3).astype(np.float32)
        tris = np.concatenate([clip, col], axis=2)
        f = oracle.Frame(W, H)
        f.clear((7, 7, 7, 7), 0.0)
        f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
        out.append((tris, f))
    return out


@pytest.mark.parametrize("tiny_lists", [False, True])
def test_bound_targets_in_flight_with_fences(oracle, monkeypatch, tiny_lists):
    """Caller-bound target sets (three, rotating), frames on the library's private streams, the caller's stream fenced after
    every frame and a copy of the frame's targets taken ON THAT STREAM (what an exchange would read) -- no warm-up frame, no
    frr_sync, no host wait until the end.  With tiny work lists the first frames overflow: the draw call itself notices and
    replays (a draw cannot fail, renderer.rs:269-384), so every copy taken through a fence is a whole frame."""
    import torch
    import f_renderer_amd as fr
    W, H = 352, 224
    sc = _scenes(oracle, W, H)
    if tiny_lists:
        monkeypatch.setenv("FRR_BIN_CAP", "3000")
    r = fr.Renderer(W, H)
    r.set_option("bound_targets_in_flight", 1)
    meshes = [r.upload_mesh(t, fr.VS_CLIP_COLOR) for t, _ in sc]
    sets = [tuple(torch.zeros((H, W), dtype=dt, device="cuda") for dt in (torch.int32, torch.float32, torch.int32)) for _ in range(3)]
    st = torch.cuda.Stream()
    taken = []

    for i in range(9):
        c_, d_, t_ = sets[i % 3]
        if not os.environ.get("FRR_TEST_SKIP_FRAME_WAIT"):   # (a manual negative control: the test must fail without the call)
            r.frame_wait(st.cuda_stream)  # the copies that still read this set (three frames back) come first
        r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(meshes[i % 2], fr.PS_COLOR)
        r.frame_fence(st.cuda_stream)
        with torch.cuda.stream(st):
            if i % 3 == 1:
                torch.cuda._sleep(30_000_000)     # a slow reader: the set's next frame (three on) must wait for these copies
            taken.append((i, c_.clone(), d_.clone(), t_.clone()))
    torch.cuda.synchronize()
    assert len(taken) == 9
    for i, c_, d_, t_ in taken:
        f = sc[i % 2][1]
        np.testing.assert_array_equal(t_.cpu().numpy().view(np.uint32).ravel(), f.tri_id, err_msg=f"frame {i}")
        np.testing.assert_array_equal(d_.cpu().numpy().view(np.uint32).ravel(), f.depth.view(np.uint32), err_msg=f"frame {i}")
        np.testing.assert_array_equal(c_.cpu().numpy().view(np.uint8).reshape(H, W, 4), f.color, err_msg=f"frame {i}")
    r.close()


def test_bound_in_flight_partial_bind_and_unbind(oracle):
    """Option bound_targets_in_flight: a partial bind is refused; going back to the ctx's own targets after an odd number of
    frames (the set index has been toggled, the second own set never allocated) works."""
    import torch
    import f_renderer_amd as fr
    W, H = 352, 224
    sc = _scenes(oracle, W, H)
    r = fr.Renderer(W, H)
    r.set_option("bound_targets_in_flight", 1)
    m = r.upload_mesh(sc[0][0], fr.VS_CLIP_COLOR)
    c_, d_, t_ = (torch.zeros((H, W), dtype=dt, device="cuda") for dt in (torch.int32, torch.float32, torch.int32))
    with pytest.raises(fr.FrrError):
        r.bind_targets(c_.data_ptr(), d_.data_ptr(), None)
    r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
    r.clear((7, 7, 7, 7), 0.0)
    r.draw(m, fr.PS_COLOR)                       # one frame: the set index is now 1
    r.bind_targets(None, None, None)             # own targets again
    r.clear((7, 7, 7, 7), 0.0)
    r.draw(m, fr.PS_COLOR)
    c, d, t = r.readback()
    f = sc[0][1]
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    np.testing.assert_array_equal(t_.cpu().numpy().view(np.uint32).ravel(), f.tri_id)   # ... and the bound frame was whole
    r.close()


class _Alias:
    """device memory at `ptr` as a torch tensor (no copy), through the CUDA array interface"""
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def test_exported_own_targets_wait_for_the_callers_reads(oracle):
    """frr_target_ptrs hands out the current frame's own target set; the caller reads it on its stream BEHIND a long-running
    kernel (a slow peer of an exchange).  Two frames later the library renders into that set again, on a private stream: it
    must wait for those reads (three different scenes in rotation, so that a torn or later frame shows)."""
    import torch
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 352, 224
    sc = _scenes(oracle, W, H)
    clip = scenes.random_clip_triangles(4000, W, H, seed=77, spread=1.2, w_jitter=0.3)
    col = scenes.splitmix_u01(78, 4000 * 9).reshape(4000, 3, 3).astype(np.float32)
    tris3 = np.concatenate([clip, col], axis=2)
    f3 = oracle.Frame(W, H)
    f3.clear((7, 7, 7, 7), 0.0)
    f3.draw(tris3, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
    sc = sc + [(tris3, f3)]
    st = torch.cuda.Stream()
    r = fr.Renderer(W, H, stream=st.cuda_stream)
    r.set_option("frames_in_flight", 2)
    meshes = [r.upload_mesh(t, fr.VS_CLIP_COLOR) for t, _ in sc]
    taken = []
    for i in range(9):
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(meshes[i % 3], fr.PS_COLOR)
        pc, pd, pt = r.target_ptrs()
        with torch.cuda.stream(st):
            torch.cuda._sleep(20_000_000)            # the caller's stream stalls; the library's private stream must not run ahead
            d = torch.as_tensor(_Alias(pd, (H, W), "<f4"), device="cuda").clone()
            t = torch.as_tensor(_Alias(pt, (H, W), "<i4"), device="cuda").clone()
            c = torch.as_tensor(_Alias(pc, (H, W), "<i4"), device="cuda").clone()
        taken.append((i, c, d, t))
    torch.cuda.synchronize()
    for i, c, d, t in taken:
        f = sc[i % 3][1]
        np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32).ravel(), f.tri_id, err_msg=f"frame {i}")
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32).ravel(), f.depth.view(np.uint32), err_msg=f"frame {i}")
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint8).reshape(H, W, 4), f.color, err_msg=f"frame {i}")
    r.close()


def test_device_bound_mesh_rewritten_in_place_between_frames(oracle):
    """A device-bound mesh whose contents the caller rewrites in place from frame to frame (animation): frr_frame_fence on the
    rewriting stream orders the rewrite behind the draws that still read the old contents, binding the mesh again orders the
    next draw behind the rewrite (include/frr.h, frr_create).  No host synchronisation; every frame is the oracle's."""
    import torch
    import f_renderer_amd as fr
    W, H = 352, 224
    sc = _scenes(oracle, W, H)
    n = min(t.shape[0] for t, _ in sc)
    frames = []
    for t, _ in sc:                                   # both scenes cut to the same triangle count (one buffer serves both)
        f = oracle.Frame(W, H)
        f.clear((7, 7, 7, 7), 0.0)
        f.draw(t[:n], oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
        frames.append(f)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        src = [torch.from_numpy(np.ascontiguousarray(t[:n])).to("cuda") for t, _ in sc]
        buf = src[0].clone()
    st.synchronize()
    r = fr.Renderer(W, H, stream=st.cuda_stream)
    r.set_option("frames_in_flight", 2)
    taken = []
    for i in range(8):
        r.frame_fence(st.cuda_stream)                 # the rewrite below follows every draw issued so far
        with torch.cuda.stream(st):
            buf.copy_(src[i % 2])
        m = r.bind_mesh_device(buf.data_ptr(), n, fr.VS_CLIP_COLOR, keepalive=buf)   # the next draw follows the rewrite
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(m, fr.PS_COLOR)
        pc, pd, pt = r.target_ptrs()
        with torch.cuda.stream(st):
            taken.append((i, torch.as_tensor(_Alias(pd, (H, W), "<f4"), device="cuda").clone(),
                          torch.as_tensor(_Alias(pt, (H, W), "<i4"), device="cuda").clone()))
    torch.cuda.synchronize()
    for i, d, t in taken:
        f = frames[i % 2]
        np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32).ravel(), f.tri_id, err_msg=f"frame {i}")
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32).ravel(), f.depth.view(np.uint32), err_msg=f"frame {i}")
    r.close()


@pytest.mark.parametrize("overlap", [0, 1, 2])
def test_own_targets_two_frames_in_flight(oracle, overlap):
    """The ctx's own targets: frames alternate between the two target sets; frr_target_ptrs hands out the current frame's
    pointers (fenced on the ctx's stream), frr_readback the current frame's image; statistics are per frame."""
    import torch
    import f_renderer_amd as fr
    W, H = 352, 224
    sc = _scenes(oracle, W, H)
    st = torch.cuda.Stream()
    r = fr.Renderer(W, H, stream=st.cuda_stream)
    r.set_option("frames_in_flight", 2)           # (the default; the suite may run with FRR_FRAMES_IN_FLIGHT=1 in the environment)
    r.set_option("overlap", overlap)
    meshes = [r.upload_mesh(t, fr.VS_CLIP_COLOR) for t, _ in sc]
    ptrs = set()
    for i in range(7):
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(meshes[i % 2], fr.PS_COLOR)
        if i >= 4:
            ptrs.add(r.target_ptrs()[1])
    c, d, t = r.readback()
    f = sc[0][1]                                  # frame 6 drew scene 0
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    st_ = r.stats()
    assert st_["draws"] == 1 and st_["tris_setup"] == f.counters.tris_setup and st_["frag_covered"] == f.counters.frag_covered
    assert len(ptrs) == 2                         # two target sets
    r.set_option("frames_in_flight", 1)
    ptrs1 = set()
    for i in range(4):
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(meshes[i % 2], fr.PS_COLOR)
        ptrs1.add(r.target_ptrs()[1])
    _, d, t = r.readback()
    np.testing.assert_array_equal(t, sc[1][1].tri_id)
    assert len(ptrs1) == 1
    r.close()
