"""Second, independent restatement of the reference's hot path in NumPy float32.

TEST INFRASTRUCTURE ONLY (like oracle/frr_oracle.c).  Written separately from the C oracle, straight
from /root/reference/f_renderer/src/renderer.rs and examples/src/bin/phong.rs, so that the two
restatements check each other bit for bit (tests/test_oracle_cross.py): the reference is Rust and
cannot be run here ("parity unpinned").  The geometry stage is a per-vertex Python
loop; the raster stage is vectorised over the bbox pixels of one triangle (each pixel is touched at
most once per triangle, so the per-pixel z rule is unaffected).  Small scenes only.
"""
import ctypes

import numpy as np

F = np.float32
_libm = ctypes.CDLL("libm.so.6")
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]

VS_CLIP, VS_CLIP_COLOR, VS_PHONG, VS_GOURAUD = 0, 1, 2, 3
PS_DEPTH, PS_FLAT, PS_COLOR, PS_PHONG, PS_BLINN = 0, 1, 2, 3, 4
VS_NF = {VS_CLIP: 4, VS_CLIP_COLOR: 7, VS_PHONG: 8, VS_GOURAUD: 8}
VS_K = {VS_CLIP: 0, VS_CLIP_COLOR: 3, VS_PHONG: 8, VS_GOURAUD: 3}
EPSILON = F(1.0e-5)                                           # renderer.rs:44
PI = F(3.14159274101257324)


def f32_as_i32(x):
    """Rust `as i32`: truncate, saturate, NaN -> 0."""
    x = float(x)
    if x != x:
        return 0
    if x >= 2147483648.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def total_key(x):
    """f32::total_cmp as an integer key."""
    i = int(np.array([x], F).view(np.int32)[0])
    if i < 0:
        i ^= 0x7FFFFFFF
    return i


class Uniforms:
    def __init__(self, model=None, view=None, proj=None, view_pos=(0, 0, 0), flat_color=(1, 1, 1, 1), tex=None):
        eye4 = np.eye(4, dtype=F).reshape(-1)
        self.model = eye4 if model is None else np.asarray(model, F).reshape(-1)
        self.view = eye4 if view is None else np.asarray(view, F).reshape(-1)
        self.proj = eye4 if proj is None else np.asarray(proj, F).reshape(-1)
        self.view_pos = np.asarray(view_pos, F)
        self.light_pos = np.array([1.2, 1.0, 2.0], F)          # phong.rs:129
        self.light_color = np.array([1.0, 1.0, 1.0], F)        # phong.rs:128
        self.ambient = F(0.1)                                  # phong.rs:131
        self.specular = F(0.5)                                 # phong.rs:132
        self.flat_color = np.asarray(flat_color, F)
        self.tex = None if tex is None else np.ascontiguousarray(tex, np.uint8)


# ---- glam pieces ------------------------------------------------------------------------------

def mat_vec(m, v):
    """Mat4 * Vec4 = ((c0*x + c1*y) + c2*z) + c3*w  (column-major)"""
    c = m.reshape(4, 4)
    return ((c[0] * v[0] + c[1] * v[1]) + c[2] * v[2]) + c[3] * v[3]


def mat_mul(a, b):
    return np.concatenate([mat_vec(a, b.reshape(4, 4)[j]) for j in range(4)]).astype(F)


def dot3(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def normalize3(a):
    r = F(1.0) / np.sqrt(dot3(a, a))
    return a * r[..., None] if np.ndim(r) else a * r


def cross3(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], F)


def set_look_at(eye, at, up):                                  # matrix_util.rs:10-22
    eye, at, up = (np.asarray(x, F) for x in (eye, at, up))
    z = normalize3(at - eye)
    x = normalize3(cross3(up, z))
    y = cross3(z, x)
    return np.array([x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0,
                     -dot3(eye, x), -dot3(eye, y), -dot3(eye, z), 1], F)


def set_perspective(fovy, aspect, zn, zf):                     # matrix_util.rs:24-35
    fovy, aspect, zn, zf = F(fovy), F(aspect), F(zn), F(zf)
    fax = F(1.0) / F(_libm.tanf(float(fovy * F(0.5))))
    m = np.zeros(16, F)
    m[0] = fax / aspect
    m[5] = fax
    m[10] = zf / (zf - zn)
    m[14] = (-zn * zf) / (zf - zn)
    m[11] = 1.0
    return m


# ---- shaders ---------------------------------------------------------------------------------

def vertex_shader(vs_id, u, vin):
    """-> (clip pos[4], ctx[K])"""
    vin = np.asarray(vin, F)
    if vs_id == VS_CLIP:
        return vin[:4].copy(), np.zeros(0, F)
    if vs_id == VS_CLIP_COLOR:
        return vin[:4].copy(), vin[4:7].copy()
    p = np.array([vin[0], vin[1], vin[2], 1.0], F)
    mvp = mat_mul(mat_mul(u.proj, u.view), u.model)            # phong.rs:119, left-associative
    world = mat_vec(u.model, p)
    if vs_id == VS_PHONG:                                      # phong.rs:114-126
        ctx = np.concatenate([vin[3:5], vin[5:8], world[:3]]).astype(F)
    else:                                                      # Gouraud: per-vertex Lambert
        n = normalize3(vin[5:8])
        l = normalize3(u.light_pos - world[:3])
        diff = max(dot3(n, l), F(0.0))
        ctx = (u.light_color * u.ambient + F(diff) * u.light_color).astype(F)
    return mat_vec(mvp, p).astype(F), ctx


def sample_2d(tex, uv):                                        # renderer.rs:516-538, vectorised
    h, w = tex.shape[0], tex.shape[1]
    x = uv[:, 0] * F(w)
    y = uv[:, 1] * F(h)
    a = x - np.trunc(x)
    b = y - np.trunc(y)

    def as_u32(v):
        v = np.where(v != v, F(0), v)
        return np.clip(np.trunc(v), 0, 4294967295.0).astype(np.int64)
    x1 = np.minimum(as_u32(x), w - 1)
    y1 = np.minimum(as_u32(y), w - 1)                          # sic: width (:523)
    x2 = np.minimum(x1 + 1, w - 1)
    y2 = np.minimum(y1 + 1, w - 1)                             # sic: width (:525)
    t = tex.astype(F) / F(255.0)
    oma, omb = F(1.0) - a, F(1.0) - b
    c11 = t[y1, x1] * oma[:, None] * omb[:, None]
    c12 = t[y2, x1] * oma[:, None] * b[:, None]
    c21 = t[y1, x2] * a[:, None] * omb[:, None]
    c22 = t[y2, x2] * a[:, None] * b[:, None]
    return ((c11 + c12) + c21) + c22


def pixel_shader(ps_id, u, ctx):
    """ctx [n,K] -> rgba [n,4] (float32).  ps_id may be a callable (u, ctx) -> rgba: a closure written for one test, the way
    the reference takes its pixel shader (renderer.rs:273,283)."""
    n = ctx.shape[0]
    if callable(ps_id):
        return np.asarray(ps_id(u, ctx), F)
    if ps_id == PS_FLAT:
        return np.broadcast_to(u.flat_color, (n, 4)).astype(F)
    if ps_id == PS_COLOR:
        return np.concatenate([ctx[:, :3], np.ones((n, 1), F)], axis=1)
    uv, normal, wpos = ctx[:, 0:2], ctx[:, 2:5], ctx[:, 5:8]
    ambient = u.light_color * u.ambient                         # phong.rs:134
    nrm = normalize3(normal)
    l = normalize3(u.light_pos - wpos)
    diff = np.maximum(dot3(nrm, l), F(0.0))
    diffuse = diff[:, None] * u.light_color
    v = normalize3(u.view_pos - wpos)
    if ps_id == PS_PHONG:
        L = -l
        t = F(2.0) * dot3(L, nrm)                               # vector_util.rs:6
        r = normalize3(t[:, None] * nrm - L)
        s = np.maximum(dot3(v, r), F(0.0))
    else:
        s = np.maximum(dot3(nrm, normalize3(l + v)), F(0.0))
    for _ in range(5):                                          # powi(32)
        s = s * s
    specular = (u.specular * s)[:, None] * u.light_color
    tex = sample_2d(u.tex, uv)
    rgb = tex[:, :3] * ((ambient + diffuse) + specular)
    return np.concatenate([rgb, (tex[:, 3] * F(1.0))[:, None]], axis=1).astype(F)


def quantize(rgba):                                            # renderer.rs:6-14
    x = rgba * F(255.0)
    x = np.where(x < 0, F(0), x)
    x = np.where(x > 255, F(255), x)
    x = np.where(x != x, F(0), x)
    return np.trunc(x).astype(np.uint8)


# ---- geometry (renderer.rs:96-267) -----------------------------------------------------------

def _inside(p):
    w = p[3]
    return [p[0] >= -w, p[0] <= w, p[1] <= w, p[1] >= -w, p[2] >= F(0.0), p[2] <= w]   # :123-131 order


def _ratio(plane, a, b):
    aw, bw = a[3], b[3]
    if plane == 0:
        return -(a[0] + aw) / (bw + b[0] - a[0] - aw)
    if plane == 1:
        return (aw - a[0]) / (aw - bw - a[0] + b[0])
    if plane == 2:
        return (aw - a[1]) / (aw - bw - a[1] + b[1])
    if plane == 3:
        return -(a[1] + aw) / (bw + b[1] - aw - a[1])
    if plane == 4:
        return aw / (aw - bw)                                    # Z_NEAR, sic (:70)
    return (aw - a[2]) / (aw - bw - a[2] + b[2])                 # Z_FAR


def geometry_processing(width, height, vs_inputs, vs_id, u):
    nf = VS_NF[vs_id]
    vin = np.asarray(vs_inputs, F).reshape(3, nf)
    verts = []
    for i in range(3):
        pos, ctx = vertex_shader(vs_id, u, vin[i])
        if pos[3] == 0.0:
            return []
        verts.append({"pos": pos.astype(F), "ctx": ctx.astype(F)})
    ins = [_inside(v["pos"]) for v in verts]
    lst = []
    if not all(all(x) for x in ins):
        with np.errstate(all="ignore"):
            for i in range(3):
                for j in range(i + 1, 3):
                    for p in range(6):
                        if ins[i][p] != ins[j][p]:
                            a, b = verts[i], verts[j]
                            t = F(_ratio(p, a["pos"], b["pos"]))
                            npos = (a["pos"] + t * (b["pos"] - a["pos"])).astype(F)
                            nctx = (a["ctx"] + (b["ctx"] - a["ctx"]) * t).astype(F)
                            if abs(npos[3]) > EPSILON:
                                lst.append({"pos": npos, "ctx": nctx})
    lst.extend(verts)
    cx = F(0.0)
    cy = F(0.0)
    for v in lst:
        cx = F(cx + v["pos"][0])
        cy = F(cy + v["pos"][1])
    inv = F(1.0) / F(len(lst))
    cx, cy = F(cx * inv), F(cy * inv)
    keys = []
    for v in lst:
        at = F(_libm.atan2f(float(F(v["pos"][1] - cy)), float(F(v["pos"][0] - cx))))
        if at < 0.0:
            at = F(at + PI * F(2.0))
        keys.append(total_key(at))
    order = sorted(range(len(lst)), key=lambda i: keys[i])      # Python's sort is stable
    lst = [lst[i] for i in order]
    with np.errstate(all="ignore"):
        for v in lst:
            rhw = F(1.0) / v["pos"][3]
            ndc = (v["pos"] * rhw).astype(F)
            sx = F(F(ndc[0] + F(1.0)) * F(width)) * F(0.5)
            sy = F(F(F(1.0) - ndc[1]) * F(height)) * F(0.5)
            v.update(rhw=F(rhw), ndc=ndc, spf=(F(sx), F(sy)), spi=(f32_as_i32(F(sx + F(0.5))), f32_as_i32(F(sy + F(0.5)))))
    n = len(lst)
    if n == 3:
        return [lst]
    tris = []
    last = n - 1
    while last > 3:
        tris.append([lst[0], lst[last - 1], lst[last]])
        last -= 1
    tris.append([lst[0], lst[2], lst[3]])
    tris.append([lst[0], lst[1], lst[2]])
    return tris


# ---- raster (renderer.rs:269-384) ------------------------------------------------------------

def _clamp(v, lo, hi):
    return lo if v < lo else (hi if v > hi else v)


def rasterization(wr, hr, tri, ps_id, u, color, depth, tri_id, tid, fb_width):
    K = tri[0]["ctx"].shape[0]
    xs = [v["spi"][0] for v in tri]
    ys = [v["spi"][1] for v in tri]
    min_x, max_x = _clamp(min(xs), wr[0], wr[1]), _clamp(max(xs), wr[0], wr[1])
    min_y, max_y = _clamp(min(ys), hr[0], hr[1]), _clamp(max(ys), hr[0], hr[1])
    if max_x <= min_x or max_y <= min_y:
        return 0
    with np.errstate(all="ignore"):
        d1 = tri[1]["ndc"] - tri[0]["ndc"]
        d2 = tri[2]["ndc"] - tri[0]["ndc"]
        nz = F(d1[0] * d2[1]) - F(d2[0] * d1[1])
    v = [tri[0], tri[2], tri[1]] if nz > 0.0 else [tri[0], tri[1], tri[2]]
    p = [np.array(x["spi"], np.int32) for x in v]

    def top_left(a, b):
        return (a[1] == b[1] and a[0] < b[0]) or a[1] > b[1]
    bias = [0 if top_left(p[0], p[1]) else 1, 0 if top_left(p[1], p[2]) else 1, 0 if top_left(p[2], p[0]) else 1]
    cy, cx = np.meshgrid(np.arange(min_y, max_y, dtype=np.int32), np.arange(min_x, max_x, dtype=np.int32), indexing="ij")
    with np.errstate(over="ignore"):
        def edge(a, b):
            return (-(cx - a[0])) * (b[1] - a[1]) + (cy - a[1]) * (b[0] - a[0])   # int32, wrapping
        cov = (edge(p[0], p[1]) >= bias[0]) & (edge(p[1], p[2]) >= bias[1]) & (edge(p[2], p[0]) >= bias[2])
    ncov = int(cov.sum())
    if ncov == 0:
        return 0
    cxs, cys = cx[cov], cy[cov]
    with np.errstate(all="ignore"):
        px, py = cxs.astype(F) + F(0.5), cys.astype(F) + F(0.5)
        s = [(F(x["spf"][0]) - px, F(x["spf"][1]) - py) for x in v]
        a = np.abs(s[1][0] * s[2][1] - s[1][1] * s[2][0])
        b = np.abs(s[2][0] * s[0][1] - s[2][1] * s[0][0])
        c = np.abs(s[0][0] * s[1][1] - s[0][1] * s[1][0])
        ssum = (a + b) + c
        ok = ssum != 0.0
        inv = F(1.0) / ssum
        a, b, c = a * inv, b * inv, c * inv
        rhw = (v[0]["rhw"] * a + v[1]["rhw"] * b) + v[2]["rhw"] * c
        idx = (cys - hr[0]).astype(np.int64) * wr[1] + (cxs - wr[0])
        passed = ok & ~(rhw < depth[idx])
        idx, rhw, a, b, c = idx[passed], rhw[passed], a[passed], b[passed], c[passed]
        depth[idx] = rhw
        tri_id[idx] = tid
        if ps_id != PS_DEPTH and idx.size:
            w = F(1.0) / np.where(rhw != 0.0, rhw, F(1.0))
            c0, c1, c2 = v[0]["rhw"] * a * w, v[1]["rhw"] * b * w, v[2]["rhw"] * c * w
            ctx = (v[0]["ctx"][None, :] * c0[:, None] + v[1]["ctx"][None, :] * c1[:, None]) + v[2]["ctx"][None, :] * c2[:, None] \
                if K else np.zeros((idx.size, 0), F)
            rgba = quantize(pixel_shader(ps_id, u, ctx.astype(F)))
            ix = (cxs - wr[0])[passed]
            iy = (cys - hr[0])[passed]
            color.reshape(-1, 4)[iy.astype(np.int64) * fb_width + ix] = rgba
    return ncov


def draw(width, height, vs_inputs, vs_id, ps_id, u, color, depth, tri_id, window=None, tri_id_base=0):
    """Loops A and B of phong.rs:319-381.  Returns (n_setup, covered fragments)."""
    nf = VS_NF[vs_id]
    vin = np.asarray(vs_inputs, F).reshape(-1, 3, nf)
    wr, hr = ((0, width), (0, height)) if window is None else ((window[0], window[1]), (window[2], window[3]))
    setup = []
    for t in range(vin.shape[0]):
        setup.extend(geometry_processing(width, height, vin[t], vs_id, u))
    cov = 0
    for i, tri in enumerate(setup):
        cov += rasterization(wr, hr, tri, ps_id, u, color, depth, tri_id, tri_id_base + i, width)
    return setup, cov
