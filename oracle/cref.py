"""ctypes binding of the C oracle (oracle/frr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package f_renderer_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfrr_oracle.so")

O_MAXK = 16
O_MAX_OUT_TRIS = 19
VS_CLIP, VS_CLIP_COLOR, VS_PHONG, VS_GOURAUD = 0, 1, 2, 3
PS_DEPTH, PS_FLAT, PS_COLOR, PS_PHONG, PS_BLINN = 0, 1, 2, 3, 4

VERTEX_DTYPE = np.dtype(
    [("ctx", "<f4", (O_MAXK,)), ("rhw", "<f4"), ("pos", "<f4", (4,)), ("spf", "<f4", (2,)), ("spi", "<i4", (2,))]
)
assert VERTEX_DTYPE.itemsize == 100


class Framebuffer(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("buffer", C.c_void_p)]


class Uniforms(C.Structure):
    _fields_ = [
        ("model", C.c_float * 16),
        ("view", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("view_pos", C.c_float * 3),
        ("light_pos", C.c_float * 3),
        ("light_color", C.c_float * 3),
        ("ambient_strength", C.c_float),
        ("specular_strength", C.c_float),
        ("flat_color", C.c_float * 4),
        ("tex", C.POINTER(Framebuffer)),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("tris_in", C.c_uint64),
        ("tris_setup", C.c_uint64),
        ("bbox_px", C.c_uint64),
        ("frag_covered", C.c_uint64),
        ("frag_zpass", C.c_uint64),
        ("frag_nan", C.c_uint64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile oracle/libfrr_oracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "frr_oracle.c")
    hdr = os.path.join(_HERE, "frr_oracle.h")
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfrr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32p, u8p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        L.o_vs_input_floats.argtypes = [C.c_int]
        L.o_vs_num_varyings.argtypes = [C.c_int]
        L.o_geometry_processing.argtypes = [C.c_uint32, C.c_uint32, f32p, C.c_int, C.POINTER(Uniforms), C.c_void_p]
        L.o_geometry_processing.restype = C.c_int
        L.o_rasterization.argtypes = [C.c_int32] * 4 + [C.c_void_p, C.c_int, C.c_int, C.POINTER(Uniforms),
                                                       C.POINTER(Framebuffer), f32p, C.c_uint64, u32p, C.c_uint32,
                                                       C.POINTER(Counters)]
        L.o_rasterization.restype = C.c_int
        L.o_draw.argtypes = [C.c_uint32, C.c_uint32] + [C.c_int32] * 4 + [f32p, C.c_uint64, C.c_int, C.c_int,
                                                                         C.POINTER(Uniforms), C.POINTER(Framebuffer),
                                                                         f32p, C.c_uint64, u32p, C.c_uint32,
                                                                         C.c_void_p, C.c_uint64, C.POINTER(Counters)]
        L.o_draw.restype = C.c_int
        L.o_geometry_batch.argtypes = [C.c_uint32, C.c_uint32, f32p, C.c_uint64, C.c_int, C.POINTER(Uniforms),
                                       C.c_void_p, C.c_uint64]
        L.o_geometry_batch.restype = C.c_int64
        L.o_fb_fill.argtypes = [C.POINTER(Framebuffer), u8p]
        L.o_depth_fill.argtypes = [f32p, C.c_uint64, C.c_float]
        L.o_sample_2d.argtypes = [C.POINTER(Framebuffer), C.c_float, C.c_float, f32p]
        L.o_sample_2d.restype = C.c_int
        L.o_vec4_to_u8.argtypes = [f32p, u8p]
        L.o_pixel_shader.argtypes = [C.c_int, C.POINTER(Uniforms), f32p, f32p]
        L.o_pixel_shader.restype = C.c_int
        L.o_set_identity.argtypes = [f32p]
        L.o_set_look_at.argtypes = [f32p, f32p, f32p, f32p]
        L.o_set_perspective.argtypes = [C.c_float] * 4 + [f32p]
        L.o_mat4_mul.argtypes = [f32p, f32p, f32p]
        L.o_mat4_mul_vec4.argtypes = [f32p, f32p, f32p]
        _lib = L
    return _lib


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vs_input_floats(vs_id):
    return lib().o_vs_input_floats(vs_id)


def vs_num_varyings(vs_id):
    return lib().o_vs_num_varyings(vs_id)


class Texture:
    """FrameBuffer used as a texture (renderer.rs:411-416)."""

    def __init__(self, rgba):
        self.data = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert self.data.ndim == 3 and self.data.shape[2] == 4
        self.fb = Framebuffer(self.data.shape[1], self.data.shape[0], self.data.ctypes.data)

    def sample_2d(self, u, v):
        out = np.zeros(4, np.float32)
        rc = lib().o_sample_2d(C.byref(self.fb), np.float32(u), np.float32(v), _f32p(out))
        return rc, out


def make_uniforms(model=None, view=None, proj=None, view_pos=(0, 0, 0), light_pos=(1.2, 1.0, 2.0),
                  light_color=(1.0, 1.0, 1.0), ambient=0.1, specular=0.5, flat_color=(1, 1, 1, 1), tex=None):
    u = Uniforms()
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    for name, m in (("model", model), ("view", view), ("proj", proj)):
        m = ident if m is None else np.asarray(m, np.float32).reshape(-1)
        getattr(u, name)[:] = [float(x) for x in m]
    u.view_pos[:] = [float(np.float32(x)) for x in view_pos]
    u.light_pos[:] = [float(np.float32(x)) for x in light_pos]
    u.light_color[:] = [float(np.float32(x)) for x in light_color]
    u.ambient_strength = float(np.float32(ambient))
    u.specular_strength = float(np.float32(specular))
    u.flat_color[:] = [float(np.float32(x)) for x in flat_color]
    if tex is not None:
        u.tex = C.pointer(tex.fb)
        u._keep = tex
    return u


def set_identity():
    m = np.zeros(16, np.float32)
    lib().o_set_identity(_f32p(m))
    return m


def set_look_at(eye, at, up):
    m = np.zeros(16, np.float32)
    e, a, u = (np.asarray(x, np.float32) for x in (eye, at, up))
    lib().o_set_look_at(_f32p(e), _f32p(a), _f32p(u), _f32p(m))
    return m


def set_perspective(fovy, aspect, zn, zf):
    m = np.zeros(16, np.float32)
    lib().o_set_perspective(np.float32(fovy), np.float32(aspect), np.float32(zn), np.float32(zf), _f32p(m))
    return m


def geometry_processing(width, height, vs_inputs, vs_id, uniforms):
    """renderer.rs:96-267 for ONE triangle -> structured array [ntri,3] (empty == None)."""
    vin = np.ascontiguousarray(vs_inputs, np.float32).reshape(-1)
    out = np.zeros((O_MAX_OUT_TRIS, 3), VERTEX_DTYPE)
    n = lib().o_geometry_processing(width, height, _f32p(vin), vs_id, C.byref(uniforms), out.ctypes.data)
    return out[:n].copy()


def geometry_batch(width, height, vs_inputs, vs_id, uniforms, cap=None):
    vin = np.ascontiguousarray(vs_inputs, np.float32)
    nf = vs_input_floats(vs_id)
    ntris = vin.size // (3 * nf)
    cap = cap or ntris * O_MAX_OUT_TRIS
    out = np.zeros((cap, 3), VERTEX_DTYPE)
    n = lib().o_geometry_batch(width, height, _f32p(vin.reshape(-1)), ntris, vs_id, C.byref(uniforms),
                               out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("setup capacity exceeded")
    return out[:n].copy()


class Frame:
    """FrameBuffer + depth buffer + (auxiliary) triangle-id buffer, as phong.rs:207-208 owns them."""

    def __init__(self, width, height, depth_len=None):
        self.width, self.height = width, height
        self.color = np.zeros((height, width, 4), np.uint8)
        self.depth = np.zeros(depth_len if depth_len is not None else width * height, np.float32)
        self.tri_id = np.full(self.depth.size, 0xFFFFFFFF, np.uint32)
        self.fb = Framebuffer(width, height, self.color.ctypes.data)
        self.counters = Counters()

    def clear(self, rgba=(30, 30, 30, 255), depth=0.0):
        c = np.asarray(rgba, np.uint8)
        lib().o_fb_fill(C.byref(self.fb), c.ctypes.data_as(C.POINTER(C.c_uint8)))
        lib().o_depth_fill(_f32p(self.depth), self.depth.size, np.float32(depth))
        self.tri_id[:] = 0xFFFFFFFF

    def rasterization(self, width_range, height_range, tri, ps_id, K, uniforms, tri_id=0):
        tri = np.ascontiguousarray(tri)
        assert tri.dtype == VERTEX_DTYPE and tri.shape == (3,)
        return lib().o_rasterization(width_range[0], width_range[1], height_range[0], height_range[1],
                                     tri.ctypes.data, ps_id, K, C.byref(uniforms), C.byref(self.fb),
                                     _f32p(self.depth), self.depth.size,
                                     self.tri_id.ctypes.data_as(C.POINTER(C.c_uint32)), tri_id,
                                     C.byref(self.counters))

    def draw(self, vs_inputs, vs_id, ps_id, uniforms, window=None, tri_id_base=0, keep_setup=False):
        vin = np.ascontiguousarray(vs_inputs, np.float32)
        nf = vs_input_floats(vs_id)
        ntris = vin.size // (3 * nf)
        x0, x1, y0, y1 = window if window is not None else (0, self.width, 0, self.height)
        setup, cap, ptr = None, 0, None
        if keep_setup:
            cap = ntris * O_MAX_OUT_TRIS
            setup = np.zeros((cap, 3), VERTEX_DTYPE)
            ptr = setup.ctypes.data
        before = int(self.counters.tris_setup)
        rc = lib().o_draw(self.width, self.height, x0, x1, y0, y1, _f32p(vin.reshape(-1)), ntris, vs_id, ps_id,
                          C.byref(uniforms), C.byref(self.fb), _f32p(self.depth), self.depth.size,
                          self.tri_id.ctypes.data_as(C.POINTER(C.c_uint32)), tri_id_base, ptr, cap,
                          C.byref(self.counters))
        if rc:
            raise RuntimeError("oracle: the reference would have panicked (bounds / clamp) or capacity exceeded")
        if keep_setup:
            return setup[: int(self.counters.tris_setup) - before].copy()
        return None


def draw_banded(width, height, vs_inputs, vs_id, ps_id, uniforms, rgba=(30, 30, 30, 255), depth=0.0, threads=None):
    """One frame with the row range split over `threads` host threads -- the reference's own sub-window
    feature (renderer.rs:270-271): every thread runs geometry + rasterization of ALL triangles into its band
    (window (0, W, y0, y1), band-local depth/colour buffers; ctypes releases the GIL).  Best-effort all-core CPU
    figure for bench.py; returns (color, depth, tri_id, covered_fragments, seconds)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    vin = np.ascontiguousarray(vs_inputs, np.float32)
    ntris = vin.size // (3 * vs_input_floats(vs_id))
    threads = max(1, min(threads or (os.cpu_count() or 1), height))
    edges = [height * i // threads for i in range(threads + 1)]
    color = np.zeros((height, width, 4), np.uint8)
    dep = np.zeros(height * width, np.float32)
    tid = np.full(height * width, 0xFFFFFFFF, np.uint32)
    counters = [Counters() for _ in range(threads)]
    L = lib()

    def band(i):
        y0, y1 = edges[i], edges[i + 1]
        if y1 <= y0:
            return
        c = color[y0:y1]                                      # contiguous row slices of the full images
        d = dep[y0 * width:y1 * width]
        t = tid[y0 * width:y1 * width]
        fb = Framebuffer(width, y1 - y0, c.ctypes.data)
        L.o_fb_fill(C.byref(fb), np.asarray(rgba, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)))
        L.o_depth_fill(_f32p(d), d.size, np.float32(depth))
        rc = L.o_draw(width, height, 0, width, y0, y1, _f32p(vin.reshape(-1)), ntris, vs_id, ps_id, C.byref(uniforms),
                      C.byref(fb), _f32p(d), d.size, t.ctypes.data_as(C.POINTER(C.c_uint32)), 0, None, 0, C.byref(counters[i]))
        if rc:
            raise RuntimeError("oracle: the reference would have panicked (bounds / clamp)")

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(band, range(threads)))
    secs = time.perf_counter() - t0
    return color, dep, tid, sum(int(c.frag_covered) for c in counters), secs
