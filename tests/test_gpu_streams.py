"""Two frames in flight: consecutive frames run on two streams with their own target sets, workspace sets and device-table
lanes (own targets: default; caller-bound targets: option bound_targets_in_flight + frr_frame_fence).  Alternating two
DIFFERENT scenes from frame to frame, without a host synchronisation in between, every frame must come out as the oracle's."""
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
    every frame and a copy of the frame's targets taken ON THAT STREAM (what an exchange would read) -- no host wait until
    the end.  With tiny work lists the first frames are replayed at the one synchronisation point in the middle."""
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

    def frame(i):
        c_, d_, t_ = sets[i % 3]
        r.bind_targets(c_.data_ptr(), d_.data_ptr(), t_.data_ptr())
        r.clear((7, 7, 7, 7), 0.0)
        r.draw(meshes[i % 2], fr.PS_COLOR)
        r.frame_fence(st.cuda_stream)
        with torch.cuda.stream(st):
            taken.append((i, c_.clone(), d_.clone(), t_.clone()))

    for i in range(2):
        frame(i)
    r.sync()                      # (work lists that were too small are grown and the frames replayed in here)
    if tiny_lists:
        taken.clear()             # copies taken through the fence before the replay saw cancelled frames: include/frr.h says so
    for i in range(2, 9):
        frame(i)
    torch.cuda.synchronize()
    assert r.stats()["replays"] == 0
    assert len(taken) >= 7
    for i, c_, d_, t_ in taken:
        f = sc[i % 2][1]
        np.testing.assert_array_equal(t_.cpu().numpy().view(np.uint32).ravel(), f.tri_id, err_msg=f"frame {i}")
        np.testing.assert_array_equal(d_.cpu().numpy().view(np.uint32).ravel(), f.depth.view(np.uint32), err_msg=f"frame {i}")
        np.testing.assert_array_equal(c_.cpu().numpy().view(np.uint8).reshape(H, W, 4), f.color, err_msg=f"frame {i}")
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
