"""The C++ host mirror (f_renderer_amd/host/frr_renderer.hpp) through the headless phong example
(examples/phong_headless.cpp = phong.rs:314-387 without the window): built with g++ against the
in-tree libfrr_hip.so, its RGBA8 frame must equal the oracle's bit for bit."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(target="phong_headless"):
    import f_renderer_amd as fr
    fr.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s", target])
    return os.path.join(ROOT, "examples", target)


def test_rccl_gather_example_compiles():
    """CPU: the native final-image exchange (examples/gather_rccl.cpp: frr_owned_rows + frr_target_ptrs + ncclSend/ncclRecv,
    one process per GPU) compiles and links against the C ABI and /opt/rocm's RCCL; without a GPU it fails loudly."""
    exe = _build("gather_rccl")
    assert os.path.exists(exe)
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe], capture_output=True, text=True)
        assert p.returncode != 0


@pytest.mark.gpu
def test_rccl_gather_example_runs_with_one_rank(tmp_path):
    """One rank (this box has one GPU): RCCL communicator, partitioned draw, the exchange group, and the comparison of the
    "gathered" image with frr_readback of an unpartitioned context -- the program exits 0 only if they are equal."""
    exe = _build("gather_rccl")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--width", "700", "--height", "333", "--tris", "30000", "--frames", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "EQUALS" in p.stdout


def test_cpp_example_compiles_against_header():
    """CPU: the C++ mirror and the example compile and link against the C ABI."""
    exe = _build()
    assert os.path.exists(exe)
    # without a GPU the example must fail loudly (no CPU fallback), not crash
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe, "/dev/null", "0", "/dev/null", "0", "64", "64", "/dev/null"], capture_output=True, text=True)
        assert p.returncode != 0


@pytest.mark.gpu
def test_cpp_phong_frame_matches_oracle(oracle, tmp_path):
    from f_renderer_amd import scenes
    exe = _build()
    W, H = 320, 180
    mesh = scenes.displaced_sphere(n=40)
    tex = scenes.checker_texture(128, 8)
    mp, tp, op, pp = (str(tmp_path / n) for n in ("mesh.f32", "tex.rgba", "out.rgba", "out.ppm"))
    mesh.tofile(mp)
    tex.tofile(tp)
    out = subprocess.run([exe, mp, str(mesh.shape[0]), tp, "128", str(W), str(H), op, pp], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(op, np.uint8).reshape(H, W, 4)
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                             view_pos=eye, tex=oracle.Texture(tex))
    f = oracle.Frame(W, H)
    f.clear((30, 30, 30, 255), 0.0)
    f.draw(mesh, oracle.VS_PHONG, oracle.PS_PHONG, u)
    np.testing.assert_array_equal(got, f.color)
    assert f"tris_setup={int(f.counters.tris_setup)}" in out.stdout
    assert open(pp, "rb").read(2) == b"P6"
