"""Host-side asset helpers: OBJ loader rules of obj_loader.rs:15-97, the BGRA texture
load semantics of renderer.rs:427-471, PPM output."""
import struct

import numpy as np
import pytest

OBJ = (
    "# comment\r\n"
    "v 0 0 0\r\n"
    "v 1 0 0\r\n"
    "v 1 1 0\r\n"
    "v 0 1 0.5\r\n"
    "vt 0 0\r\n"
    "vt 1 0\r\n"
    "vt 1 1\r\n"
    "vn 0 0 2\r\n"
    "vn 3 0 4\r\n"
    "g ignored group\r\n"
    "f 1/1/1 2/2/1 3/3/2 4/1/1\r\n"      # a quad: only the first three triples are used (:59)
    "f 1/1/2 3/3/2 4/2/1\r\n"
)


def test_obj_loader_rules():
    from f_renderer_amd.assets import Model
    m = Model(data=OBJ.encode())
    assert (len(m.verts), len(m.uv), len(m.norms), m.faces_len()) == (4, 3, 2, 2)
    assert m.faces[0] == [(0, 0, 0), (1, 1, 0), (2, 2, 1)]           # 1-based -> 0-based
    assert m.vert(1, 2).tolist() == [0.0, 1.0, 0.5]
    assert m.uv_at(0, 1).tolist() == [1.0, 0.0]
    np.testing.assert_array_equal(m.normal(0, 0), np.array([0, 0, 1], np.float32))       # normalised at fetch
    n = m.normal(0, 2)
    assert abs(float(n[0]) - 0.6) < 1e-6 and abs(float(n[2]) - 0.8) < 1e-6
    vin = m.vertex_inputs()
    assert vin.shape == (2, 3, 8) and vin.dtype == np.float32
    assert vin[0, 1].tolist() == [1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]


def test_obj_loader_panics_become_errors():
    from f_renderer_amd.assets import Model
    with pytest.raises(ValueError):
        Model(data=b"v 1 2 x\n")                  # parse::<f32>().unwrap()
    with pytest.raises(ValueError):
        Model(data=b"v 0 0 0\nvt 0 0\nvn 0 0 1\nf 0/1/1 1/1/1 1/1/1\n")   # 0 - 1 underflows u32
    with pytest.raises(IndexError):
        Model(data=b"v 1 2\n")                    # l_v[3] out of bounds


def test_texture_bgra_semantics_and_tga(tmp_path):
    from f_renderer_amd.assets import load_tga, save_ppm, texture_from_image
    rgb = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    fb = texture_from_image(rgb)
    assert fb.get_pixel(1, 0).tolist() == [5, 4, 3, 255]             # B,G,R,255 (renderer.rs:442-445)
    rgba = np.dstack([rgb, np.full((2, 3), 7, np.uint8)])
    assert texture_from_image(rgba).get_pixel(2, 1).tolist() == [17, 16, 15, 7]
    with pytest.raises(ValueError):
        texture_from_image(np.zeros((2, 2), np.uint8))
    # uncompressed 24-bit TGA, bottom-left origin: rows stored bottom-up, pixels B,G,R
    p = tmp_path / "t.tga"
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 3, 2, 24, 0)
    p.write_bytes(hdr + rgb[::-1, :, ::-1].tobytes())
    np.testing.assert_array_equal(load_tga(str(p)).buffer, fb.buffer)
    # PPM writer
    out = tmp_path / "o.ppm"
    save_ppm(fb, str(out), bgra=True)
    data = out.read_bytes()
    assert data.startswith(b"P6\n3 2\n255\n") and data[-18:] == rgb.tobytes()
