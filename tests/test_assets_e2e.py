"""f2/f3 end to end: the reference's own asset formats -- OBJ through Model::new's parsing rules
(obj_loader.rs:15-97) and init_vertex_input (phong.rs:187-201), TGA through FrameBuffer::load_file's BGRA storage
(renderer.rs:427-471) -- feed the HIP path, and the frame equals the oracle's bit for bit.  Fixtures (data):
tests/golden/cube.obj (hand-written: CRLF, a quad face, consecutive spaces, signed / exponent literals, unnormalised
normals), checker24.tga, checker32_rle.tga, run24_rle.tga (tests/golden/make_assets.py).  The compiled host's
loaders (frr::Model, frr::FrameBuffer::load_file in frr_renderer.hpp) must produce the same bytes as the Python ones."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
OBJ = os.path.join(G, "cube.obj")
TGAS = ["checker24.tga", "checker32_rle.tga", "run24_rle.tga"]


def _exe():
    import f_renderer_amd as fr
    fr.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s", "phong_headless"])
    return os.path.join(ROOT, "examples", "phong_headless")


def test_fixture_content():
    from f_renderer_amd.assets import Model, load_tga
    raw = open(OBJ, "rb").read()
    assert b"\r\n" in raw and b"  " in raw and b"6e-1" in raw and b"+0.6" in raw
    m = Model(OBJ)
    assert (len(m.verts), len(m.uv), len(m.norms), m.faces_len()) == (8, 4, 6, 12)
    assert m.faces[0] == [(0, 0, 0), (1, 1, 0), (2, 2, 0)]                   # the quad line: first three triples only
    np.testing.assert_array_equal(m.normal(0, 0), np.array([0, 0, 1], np.float32))   # (0,0,2) normalised at fetch
    a, b, c = (load_tga(os.path.join(G, t)) for t in TGAS)
    assert a.buffer[0, 0].tolist() == [64, 240, 16, 255]                      # B,G,R,255: top-left pixel of a bottom-up file
    np.testing.assert_array_equal(a.buffer[..., :3], b.buffer[..., :3])      # same picture through both encodings
    assert b.buffer[7, 7].tolist() == [64, 72, 212, 255 - 56 - 28]
    assert c.buffer[0].reshape(-1, 4).tolist() == [[10, 200, 90, 255]] * 8    # the run packet
    np.testing.assert_array_equal(c.buffer[1:], a.buffer[1:])


@pytest.mark.parametrize("tga", TGAS)
def test_compiled_host_loaders_match_python(tmp_path, tga):
    """CPU: frr::Model / frr::FrameBuffer::load_file (C++) against f_renderer_amd.assets (Python), byte for byte."""
    from f_renderer_amd.assets import Model, load_tga
    exe = _exe()
    mo, to = str(tmp_path / "m.f32"), str(tmp_path / "t.rgba")
    out = subprocess.run([exe, "--dump-assets", OBJ, os.path.join(G, tga), mo, to], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    vin = Model(OBJ).vertex_inputs()
    np.testing.assert_array_equal(np.fromfile(mo, np.float32).view(np.uint32), vin.reshape(-1).view(np.uint32))
    np.testing.assert_array_equal(np.fromfile(to, np.uint8), load_tga(os.path.join(G, tga)).buffer.reshape(-1))


def test_compiled_host_loader_errors(tmp_path):
    """what panics in the reference (parse unwrap, index, u32 underflow, unknown image type) is an frr::Error"""
    exe = _exe()
    bad = tmp_path / "bad.obj"
    for text in ("v 1 2 x\n", "v 1 2\n", "v 0 0 0\nvt 0 0\nvn 0 0 1\nf 0/1/1 1/1/1 1/1/1\n", "v 0  0 0\n"):
        bad.write_text(text)
        p = subprocess.run([exe, "--dump-assets", str(bad), os.path.join(G, TGAS[0]), os.devnull, os.devnull], capture_output=True, text=True)
        assert p.returncode == 1 and "frr error" in p.stderr, (text, p.stderr)
    p = subprocess.run([exe, "--dump-assets", OBJ, OBJ, os.devnull, os.devnull], capture_output=True, text=True)
    assert p.returncode == 1


def _oracle_frame(oracle, vin, tex, W, H, ps):
    from f_renderer_amd import scenes
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                             view_pos=eye, tex=oracle.Texture(tex))
    f = oracle.Frame(W, H)
    f.clear((30, 30, 30, 255), 0.0)
    f.draw(vin, oracle.VS_PHONG, getattr(oracle, "PS_" + ps), u)
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("tga", TGAS)
@pytest.mark.parametrize("ps", ["PHONG", "BLINN"])
def test_obj_tga_through_hip_path(oracle, tga, ps):
    """Model -> vertex_inputs -> frr_draw with the BGRA texture == oracle (depth bits, ids, RGBA8)."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    from f_renderer_amd.assets import Model, load_tga
    W, H = 384, 216
    vin = Model(OBJ).vertex_inputs()
    tex = load_tga(os.path.join(G, tga))
    r = fr.Renderer(W, H)
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    r.set_texture(0, tex)
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    r.clear((30, 30, 30, 255), 0.0)
    r.draw(r.upload_mesh(vin, fr.VS_PHONG), getattr(fr, "PS_" + ps))
    c, d, t = r.readback()
    f = _oracle_frame(oracle, vin, tex.buffer, W, H, ps)
    assert (t != 0xFFFFFFFF).sum() > 5000                                     # the cube is on screen
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))
    np.testing.assert_array_equal(c, f.color)
    assert r.stats()["tris_setup"] == f.counters.tris_setup == 12


@pytest.mark.gpu
def test_cpp_example_renders_obj_and_tga(oracle, tmp_path):
    """examples/phong_headless --assets: the compiled host loads the OBJ + TGA itself and its frame equals the oracle's."""
    from f_renderer_amd.assets import Model, load_tga
    exe = _exe()
    W, H = 320, 180
    op = str(tmp_path / "out.rgba")
    tga = os.path.join(G, "checker32_rle.tga")
    out = subprocess.run([exe, "--assets", OBJ, tga, str(W), str(H), op], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(op, np.uint8).reshape(H, W, 4)
    f = _oracle_frame(oracle, Model(OBJ).vertex_inputs(), load_tga(tga).buffer, W, H, "PHONG")
    np.testing.assert_array_equal(got, f.color)
    assert "tris_setup=12" in out.stdout
