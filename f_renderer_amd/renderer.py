"""Host-side mirror of the reference's rasterization interface over the C ABI (include/frr.h).

Reference surface (Rust, /root/reference/f_renderer/src/renderer.rs):
  Renderer::geometry_processing(width, height, vs_inputs, vertex_shader, vs_uniform)   :96-112
  Renderer::rasterization(width_range, height_range, triangle, pixel_shader, ps_uniform,
                          frame_buffer, depth_buffer)                                  :269-284
  FrameBuffer::{new, fill, clear, get_size, get_data, set_pixel, get_pixel}            :418-514
and the matrix/camera helpers matrix_util.rs:3-35, camera.rs:4-26.

The per-triangle, closure-taking calls become batched calls with table-selected shaders: same
names, same argument meaning (width_range/height_range tuples, clear colours, uniforms), same
error behaviour (what panics in the reference raises FrrError(FRR_ERR_INVALID) here).
This module is plumbing only: all arithmetic of the path runs in libfrr_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import (FrrError, PS_BLINN, PS_COLOR, PS_DEPTH, PS_FLAT, PS_PHONG, VS_CLIP, VS_CLIP_COLOR,  # noqa: F401
                      VS_GOURAUD, VS_PHONG)

SETUP_DTYPE = np.dtype([("spf", "<f4", (2,)), ("spi", "<i4", (2,)), ("rhw", "<f4"), ("ctx", "<f4", (N.MAX_VARYINGS,))])
assert SETUP_DTYPE.itemsize == C.sizeof(N.SetupVertex)

_f32p = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(_f32p)


# ---- matrix_util.rs / camera.rs -------------------------------------------------------------

def set_identity():
    """matrix_util.rs:3-8"""
    m = np.zeros(16, np.float32)
    N.lib().frr_set_identity(_fp(m))
    return m


def set_look_at(eye, at, up):
    """matrix_util.rs:10-22 (left-handed look-at, column-major)"""
    m = np.zeros(16, np.float32)
    e, a, u = (np.ascontiguousarray(x, np.float32) for x in (eye, at, up))
    N.lib().frr_set_look_at(_fp(e), _fp(a), _fp(u), _fp(m))
    return m


def set_perspective(fovy, aspect, zn, zf):
    """matrix_util.rs:24-35 (LH, depth 0..1, w_clip = z_view)"""
    m = np.zeros(16, np.float32)
    N.lib().frr_set_perspective(np.float32(fovy), np.float32(aspect), np.float32(zn), np.float32(zf), _fp(m))
    return m


class Camera:
    """camera.rs:4-26"""

    def __init__(self, eye, at, up):
        self.eye, self.at, self.up = (np.asarray(x, np.float32) for x in (eye, at, up))
        self.mat_look_at = set_look_at(self.eye, self.at, self.up)

    def cal_look_at(self):
        self.mat_look_at = set_look_at(self.eye, self.at, self.up)
        return self.mat_look_at


# ---- FrameBuffer ------------------------------------------------------------------------------

class FrameBuffer:
    """Host image with the reference's FrameBuffer surface (renderer.rs:411-514): RGBA8 row-major,
    offset (y*width + x)*4.  Used for textures and for what Renderer.frame_buffer() reads back."""

    def __init__(self, width, height, data=None):
        self.width, self.height = int(width), int(height)
        self.buffer = np.zeros((self.height, self.width, 4), np.uint8) if data is None else \
            np.ascontiguousarray(data, np.uint8).reshape(self.height, self.width, 4)

    @staticmethod
    def new(width, height):
        return FrameBuffer(width, height)

    def get_data(self):
        return self.buffer.reshape(-1)

    def get_size(self):
        return self.width * self.height * 4

    def clear(self):
        self.buffer[...] = 0

    def fill(self, color):
        self.buffer[...] = np.asarray(color, np.uint8)

    def set_pixel(self, x, y, color):
        self.buffer[y, x] = np.asarray(color, np.uint8)

    def get_pixel(self, x, y):
        return self.buffer[y, x].copy()


class Mesh:
    def __init__(self, renderer, mesh_id, ntris, vs_id, keepalive=None):
        self.renderer, self.id, self.ntris, self.vs_id, self._keep = renderer, mesh_id, ntris, vs_id, keepalive

    def free(self):
        if self.id is not None:
            self.renderer._check(N.lib().frr_mesh_free(self.renderer._ctx, self.id))
            self.id = None


def _options_from_env():
    """The test-suite and the tools select code paths with FRR_* environment variables; the library reads none, so they
    are translated into frr_set_option calls here (every new Renderer)."""
    import os
    e, o = os.environ, {}
    if e.get("FRR_RASTER") == "sweep":
        o["raster_sweep"] = 1
    for var, name in (("FRR_RASTER_NW", "raster_nw"), ("FRR_RASTER_OCC", "raster_occ"), ("FRR_BIN_G", "bin_chunks"),
                      ("FRR_ENT_SLOT", "tile_slot_records"), ("FRR_BIN_CAP", "bin_capacity"), ("FRR_FAN_CAP", "fan_capacity"),
                      ("FRR_CLIP_QUEUE", "clip_queue"), ("FRR_OVERLAP", "overlap"), ("FRR_FRAMES_IN_FLIGHT", "frames_in_flight"),
                      ("FRR_BOUND_IN_FLIGHT", "bound_targets_in_flight")):
        if e.get(var):
            o[name] = int(e[var])
    if e.get("FRR_CLEAR") == "eager":
        o["clear_eager"] = 1
    if e.get("FRR_BIN") == "atomics":
        o["bin_atomics"] = 1
    return o


class Renderer:
    """Device-resident FrameBuffer (width x height RGBA8) + f32 depth buffer + u32 triangle-id
    buffer, and the two halves of the reference's draw loop (phong.rs:319-381) as batched calls."""

    def __init__(self, width, height, device=0, stream=None):
        self._lib = N.lib()
        self.width, self.height = int(width), int(height)
        ctx = C.c_void_p()
        rc = self._lib.frr_create(int(device), self.width, self.height, C.c_void_p(stream or 0), C.byref(ctx))
        if rc != N.FRR_OK:
            raise FrrError(rc, "frr_create failed (no gfx950 device, bad size, or out of memory); "
                               "there is no CPU fallback")
        self._ctx = ctx
        self.uniforms = N.Uniforms()
        ident = [1.0 if i % 5 == 0 else 0.0 for i in range(16)]
        self.uniforms.model[:] = ident
        self.uniforms.view[:] = ident
        self.uniforms.proj[:] = ident
        self.uniforms.light_pos[:] = [float(np.float32(1.2)), 1.0, 2.0]
        self.uniforms.light_color[:] = [1.0, 1.0, 1.0]
        self.uniforms.ambient_strength = float(np.float32(0.1))
        self.uniforms.specular_strength = 0.5
        self.uniforms.flat_color[:] = [1.0, 1.0, 1.0, 1.0]
        self._keep = []
        self.last_warning = None
        for name, value in _options_from_env().items():
            self.set_option(name, value)

    def set_option(self, name, value):
        """Development / test switches of the library (include/frr.h: frr_set_option); none changes a result."""
        self._check(self._lib.frr_set_option(self._ctx, name.encode(), int(value)))

    # -- lifetime / errors --
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.frr_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        """Negative status: raise.  Positive status (a warning; none is defined at present): the call delivered its results; remembered in
        `last_warning` (None when the latest checked call had nothing to report)."""
        if rc < N.FRR_OK:
            raise FrrError(rc, self._lib.frr_last_error(self._ctx).decode())
        self.last_warning = rc if rc > N.FRR_OK else None

    # -- scene --
    def upload_mesh(self, vs_inputs, vs_id):
        """Vec<[VSInput;3]> (phong.rs:187-205) -> device.  vs_inputs: float32 [ntris,3,NF]."""
        nf = self._lib.frr_vs_input_floats(vs_id)
        if nf < 0:
            raise FrrError(N.FRR_ERR_INVALID, "unknown vertex shader id")
        a = np.ascontiguousarray(vs_inputs, np.float32)
        if a.size % (3 * nf):
            raise FrrError(N.FRR_ERR_INVALID, "vs_inputs size is not a multiple of 3*floats_per_vertex")
        ntris = a.size // (3 * nf)
        mid = C.c_int()
        self._check(self._lib.frr_mesh_upload(self._ctx, a.ctypes.data, ntris, vs_id, C.byref(mid)))
        return Mesh(self, mid.value, ntris, vs_id)

    def bind_mesh_device(self, dev_ptr, ntris, vs_id, keepalive=None):
        """Same, for data already in HBM (e.g. a torch tensor's data_ptr())."""
        mid = C.c_int()
        self._check(self._lib.frr_mesh_bind_device(self._ctx, C.c_void_p(dev_ptr), ntris, vs_id, C.byref(mid)))
        return Mesh(self, mid.value, ntris, vs_id, keepalive)

    def set_texture(self, slot, image):
        """PSUniform.sample_2d_* (phong.rs:43-45): FrameBuffer or uint8 [h,w,4]."""
        buf = image.buffer if isinstance(image, FrameBuffer) else np.ascontiguousarray(image, np.uint8)
        h, w = buf.shape[0], buf.shape[1]
        self._check(self._lib.frr_texture_upload(self._ctx, slot, buf.ctypes.data, w, h))

    def set_uniforms(self, model=None, view=None, proj=None, view_pos=None, light_pos=None, light_color=None,
                     ambient_strength=None, specular_strength=None, flat_color=None, texture_slot=None):
        u = self.uniforms
        for name, v in (("model", model), ("view", view), ("proj", proj), ("view_pos", view_pos),
                        ("light_pos", light_pos), ("light_color", light_color), ("flat_color", flat_color)):
            if v is not None:
                getattr(u, name)[:] = [float(x) for x in np.asarray(v, np.float32).reshape(-1)]
        if ambient_strength is not None:
            u.ambient_strength = float(np.float32(ambient_strength))
        if specular_strength is not None:
            u.specular_strength = float(np.float32(specular_strength))
        if texture_slot is not None:
            u.texture_slot = int(texture_slot)
        self._check(self._lib.frr_set_uniforms(self._ctx, C.byref(u)))

    def register_shader(self, hip_source, vs_input_floats, num_varyings):
        """The reference's closure API (renderer.rs:105,283) as text: HIP source defining frr_user_vs / frr_user_ps
        (include/frr.h: user shaders), compiled at run time into the library's kernels.  Returns the shader id to pass as
        vs_id to upload_mesh and as pixel_shader to rasterization / draw."""
        sid = C.c_int()
        self._check(self._lib.frr_shader_register(self._ctx, hip_source.encode(), int(vs_input_floats), int(num_varyings), C.byref(sid)))
        return sid.value

    def set_user_uniforms(self, values):
        """u.user[...] of the user shaders: what the reference's closures would have captured."""
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        self._check(self._lib.frr_set_user_uniforms(self._ctx, _fp(v), v.size))

    def set_partition(self, rank, world, blocked=False):
        """Tile-row ownership of a multi-GPU rank: interleaved rows (ty % world == rank) or, blocked=True,
        a contiguous run of tile_rows // world rows (the first tile_rows % world ranks one more)."""
        self._check(self._lib.frr_set_partition(self._ctx, rank, world))
        self._check(self._lib.frr_set_partition_layout(self._ctx, 1 if blocked else 0))

    def owned_rows(self, height_range=None):
        """Bands [row0, row1) of window-local pixel rows this rank owns (frr_owned_rows): what the final-image
        gather of a multi-GPU run sends."""
        y0, y1 = height_range or (0, self.height)
        n = self._lib.frr_owned_band_count(self._ctx, y0, y1)
        if n < 0:
            raise FrrError(n, "frr_owned_band_count")
        out = []
        for b in range(n):
            a, e = C.c_int32(), C.c_int32()
            self._check(self._lib.frr_owned_rows(self._ctx, y0, y1, b, C.byref(a), C.byref(e)))
            out.append((a.value, e.value))
        return out

    def set_count_fragments(self, enable):
        self._check(self._lib.frr_set_count_fragments(self._ctx, 1 if enable else 0))

    def bind_targets(self, color_ptr=None, depth_ptr=None, tri_id_ptr=None):
        self._check(self._lib.frr_bind_targets(self._ctx, C.c_void_p(color_ptr or 0), C.c_void_p(depth_ptr or 0),
                                               C.c_void_p(tri_id_ptr or 0)))

    def target_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._lib.frr_target_ptrs(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- frame --
    def clear(self, color=(30, 30, 30, 255), depth=0.0):
        """frame_buffer.fill(color); depth_buffer.fill(depth)  (phong.rs:316-317)"""
        c = np.asarray(color, np.uint8)
        self._check(self._lib.frr_clear(self._ctx, c.ctypes.data, np.float32(depth)))

    def geometry_processing(self, mesh, count=False):
        """Loop A (phong.rs:321-331): Renderer::geometry_processing over all triangles of mesh."""
        n = C.c_uint64()
        self._check(self._lib.frr_geometry(self._ctx, mesh.id, C.byref(n) if count else None))
        return int(n.value) if count else None

    def rasterization(self, width_range, height_range, pixel_shader):
        """Loop B (phong.rs:361-381): Renderer::rasterization of the last geometry_processing."""
        self._check(self._lib.frr_raster(self._ctx, pixel_shader, int(width_range[0]), int(width_range[1]),
                                         int(height_range[0]), int(height_range[1])))

    def draw(self, mesh, pixel_shader, width_range=None, height_range=None):
        wr = width_range or (0, self.width)
        hr = height_range or (0, self.height)
        self._check(self._lib.frr_draw(self._ctx, mesh.id, pixel_shader, wr[0], wr[1], hr[0], hr[1]))

    def frame_fence(self, stream=None):
        """`stream` (a hipStream_t handle, e.g. torch's stream.cuda_stream; None = the ctx's stream) waits for every frame
        issued so far -- no host wait (frr_frame_fence; option bound_targets_in_flight)."""
        self._check(self._lib.frr_frame_fence(self._ctx, C.c_void_p(stream or 0)))

    def frame_wait(self, stream=None):
        """The next kernel that writes the frame targets waits for what `stream` (default: the ctx's) holds now
        (frr_frame_wait): e.g. an exchange that still reads a target set about to be bound for a new frame."""
        self._check(self._lib.frr_frame_wait(self._ctx, C.c_void_p(stream or 0)))

    def sync(self):
        self._check(self._lib.frr_sync(self._ctx))

    # -- results --
    def readback(self, color=True, depth=True, tri_id=True):
        n = self.width * self.height
        c = np.empty((self.height, self.width, 4), np.uint8) if color else None
        d = np.empty(n, np.float32) if depth else None
        t = np.empty(n, np.uint32) if tri_id else None
        self._check(self._lib.frr_readback(self._ctx, c.ctypes.data if color else None, d.ctypes.data if depth else None,
                                           t.ctypes.data if tri_id else None))
        return c, d, t

    def frame_buffer(self):
        """FrameBuffer with get_data() as phong.rs:386 reads it."""
        c, _, _ = self.readback(True, False, False)
        return FrameBuffer(self.width, self.height, c)

    def setup_triangles(self):
        """Vec<[Vertex;3]> of the last geometry_processing (emission order), for parity tests."""
        n = C.c_uint64()
        self._check(self._lib.frr_readback_setup(self._ctx, None, 0, C.byref(n)))
        out = np.zeros((int(n.value), 3), SETUP_DTYPE)
        if n.value:
            self._check(self._lib.frr_readback_setup(self._ctx, out.ctypes.data, n.value, C.byref(n)))
        return out

    def stats(self):
        s = N.Stats()
        self._check(self._lib.frr_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    # -- timing --
    def event_record(self, slot):
        self._check(self._lib.frr_event_record(self._ctx, slot))

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self._check(self._lib.frr_event_elapsed_ms(self._ctx, a, b, C.byref(ms)))
        return float(ms.value)

    KERNELS = ("k_clear", "k_geom", "k_geom_scan", "k_bin_count",
               "k_tile_scan", "k_bin_fill", "k_raster", "k_bin_seg")

    def profile_enable(self, on=True, kernels=None, period=1):
        """Bracket launches with HIP events: all kernels (on=True), none (False) or the named ones; only every
        `period`-th launch of each (an event pair costs the stream ~4 us)."""
        mask = (-1 if on else 0) if kernels is None else sum(1 << self.KERNELS.index(k) for k in kernels)
        self._check(self._lib.frr_profile_set_period(self._ctx, period))
        self._check(self._lib.frr_profile_enable(self._ctx, mask))

    def profile_reset(self):
        self._check(self._lib.frr_profile_reset(self._ctx))

    def profile_get(self, kernel):
        ms, n = C.c_float(), C.c_uint32()
        self._check(self._lib.frr_profile_get(self._ctx, kernel.encode(), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def debug_rcp_check(self, lo_bits, hi_bits):
        """(mismatches, first offending bit pattern) of recip_exact vs the IEEE division on [lo_bits, hi_bits)."""
        n, first = C.c_uint64(0), C.c_uint32(0)
        self._check(self._lib.frr_debug_rcp_check(self._ctx, lo_bits, hi_bits, C.byref(n), C.byref(first)))
        return int(n.value), int(first.value)

    def debug_scan64(self, values):
        v = np.ascontiguousarray(values, np.uint32)
        assert v.size == 64
        out = np.empty(64, np.uint32)
        self._check(self._lib.frr_debug_scan64(self._ctx, v.ctypes.data, out.ctypes.data))
        return out

    def debug_atan2f(self, y, x):
        y = np.ascontiguousarray(y, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(y)
        self._check(self._lib.frr_debug_atan2f(self._ctx, y.ctypes.data, x.ctypes.data, out.ctypes.data, y.size))
        return out
