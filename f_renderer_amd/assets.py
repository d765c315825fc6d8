"""Host-side asset helpers on the edges of the hot path.  Plain I/O, no compute
path: they produce the `Vec<[VSInput;3]>` arrays and RGBA8 textures the C ABI takes.

* `Model`            -- OBJ loader with the reference's exact parsing rules
                        (/root/reference/f_renderer/src/obj_loader.rs:15-97) and
                        `vertex_inputs()` = `init_vertex_input` of examples/src/bin/phong.rs:187-201;
* `texture_from_image` / `load_tga` -- `FrameBuffer::load_file` semantics (renderer.rs:427-471):
                        RGB/RGBA decoded top-down, stored **BGRA**, alpha 255 for RGB sources;
* `save_ppm`         -- headless stand-in for the Vulkan/wgpu presentation (phong.rs:386).
"""
import struct

import numpy as np

from .renderer import FrameBuffer

F = np.float32


def _parse_f32(tok):
    """Rust `str::parse::<f32>()` after `.replace("\\r", "")`; `.unwrap()` panics -> ValueError here."""
    tok = tok.replace("\r", "")
    if tok != tok.strip() or "_" in tok or tok == "":
        raise ValueError(f"invalid f32 literal {tok!r} (the reference would panic)")
    return F(float(tok))


def _parse_u32_minus_1(tok):
    tok = tok.replace("\r", "")
    if not tok.isdigit() and not (tok.startswith("+") and tok[1:].isdigit()):
        raise ValueError(f"invalid u32 literal {tok!r} (the reference would panic)")
    v = int(tok)
    if v == 0 or v > 0xFFFFFFFF:
        raise ValueError("index 0 underflows `u32 - 1` / overflows u32 (the reference would panic)")
    return v - 1


class Model:
    """obj_loader.rs:7-97.  Lines are split on "\\n", tokens on single spaces (so consecutive spaces
    yield empty tokens, as in the reference); only `v`, `vn`, `vt`, `f` lines are used; a face takes
    its first three `a/b/c` triples, 1-based -> 0-based; normals are normalised when fetched."""

    def __init__(self, path=None, data=None):
        if data is None:
            with open(path, "rb") as fh:
                data = fh.read()
        text = data.decode("utf-8", errors="replace")                  # String::from_utf8_lossy (:28)
        self.verts, self.norms, self.uv, self.faces = [], [], [], []
        for line in text.split("\n"):                                  # :29
            l_v = line.split(" ")                                      # :32
            tag = l_v[0]
            if tag == "v":
                self.verts.append([_parse_f32(l_v[1]), _parse_f32(l_v[2]), _parse_f32(l_v[3])])       # :37-43
            elif tag == "vn":
                self.norms.append([_parse_f32(l_v[1]), _parse_f32(l_v[2]), _parse_f32(l_v[3])])       # :44-50
            elif tag == "vt":
                self.uv.append([_parse_f32(l_v[1]), _parse_f32(l_v[2])])                              # :51-56
            elif tag == "f":
                tri = []
                for i in range(1, 4):                                                                 # :59-67
                    vv = l_v[i].split("/")
                    tri.append((_parse_u32_minus_1(vv[0]), _parse_u32_minus_1(vv[1]), _parse_u32_minus_1(vv[2])))
                self.faces.append(tri)

    def faces_len(self):
        return len(self.faces)

    def vert(self, i_face, nth_vert):
        return np.array(self.verts[self.faces[i_face][nth_vert][0]], F)

    def uv_at(self, i_face, nth_vert):                                  # `uv()` in the reference (:86-89)
        return np.array(self.uv[self.faces[i_face][nth_vert][1]], F)

    def normal(self, i_face, nth_vert):
        n = np.array(self.norms[self.faces[i_face][nth_vert][2]], F)
        dot = F(F(n[0] * n[0]) + F(n[1] * n[1])) + F(n[2] * n[2])      # glam Vec3::normalize (:93-96)
        with np.errstate(all="ignore"):
            return (n * (F(1.0) / np.sqrt(dot))).astype(F)

    def vertex_inputs(self):
        """phong.rs:187-201: float32 [faces, 3, 8] = (pos3, uv2, normal3), the FRR_VS_PHONG layout."""
        out = np.zeros((self.faces_len(), 3, 8), F)
        for i in range(self.faces_len()):
            for j in range(3):
                out[i, j, 0:3] = self.vert(i, j)
                out[i, j, 3:5] = self.uv_at(i, j)
                out[i, j, 5:8] = self.normal(i, j)
        return out


def texture_from_image(pixels):
    """renderer.rs:435-464: uint8 [h,w,3] (Rgb8) or [h,w,4] (Rgba8), top row first -> FrameBuffer
    holding B,G,R,A (alpha 255 for Rgb8).  Other layouts panic in the reference -> ValueError."""
    px = np.asarray(pixels)
    if px.dtype != np.uint8 or px.ndim != 3 or px.shape[2] not in (3, 4):
        raise ValueError("invalid color type (the reference panics, renderer.rs:461-463)")
    h, w, c = px.shape
    out = np.empty((h, w, 4), np.uint8)
    out[..., 0] = px[..., 2]
    out[..., 1] = px[..., 1]
    out[..., 2] = px[..., 0]
    out[..., 3] = 255 if c == 3 else px[..., 3]
    return FrameBuffer(w, h, out)


def load_tga(path):
    """Decode an uncompressed or RLE true-colour TGA (types 2 / 10, 24 or 32 bpp) to top-down RGB(A)
    and store it as FrameBuffer::load_file does (the reference decodes with the `image` crate)."""
    with open(path, "rb") as fh:
        d = fh.read()
    idlen, cmap, typ = d[0], d[1], d[2]
    w, h, bpp, desc = struct.unpack_from("<HHBB", d, 12)
    if cmap != 0 or typ not in (2, 10) or bpp not in (24, 32):
        raise ValueError("unsupported TGA (only true-colour 24/32 bpp, types 2 and 10)")
    n, bp, pos = w * h, bpp // 8, 18 + idlen
    if typ == 2:
        raw = np.frombuffer(d, np.uint8, n * bp, pos).reshape(n, bp)
    else:
        buf, i = np.empty((n, bp), np.uint8), 0
        while i < n:
            hdr = d[pos]; pos += 1
            cnt = (hdr & 0x7F) + 1
            if hdr & 0x80:
                buf[i:i + cnt] = np.frombuffer(d, np.uint8, bp, pos); pos += bp
            else:
                buf[i:i + cnt] = np.frombuffer(d, np.uint8, cnt * bp, pos).reshape(cnt, bp); pos += cnt * bp
            i += cnt
        raw = buf
    img = raw.reshape(h, w, bp)[..., [2, 1, 0] + ([3] if bp == 4 else [])]   # TGA stores B,G,R(,A)
    if not (desc & 0x20):
        img = img[::-1]                                                      # bottom-left origin -> top-down
    if desc & 0x10:
        img = img[:, ::-1]
    return texture_from_image(np.ascontiguousarray(img))


def save_ppm(frame_buffer, path, bgra=False):
    """Binary PPM of a FrameBuffer's RGB bytes (`bgra=True` if its bytes are B,G,R,A as presented by the
    reference's swapchain)."""
    px = frame_buffer.buffer[..., :3]
    if bgra:
        px = px[..., ::-1]
    with open(path, "wb") as fh:
        fh.write(f"P6\n{frame_buffer.width} {frame_buffer.height}\n255\n".encode())
        fh.write(np.ascontiguousarray(px).tobytes())
