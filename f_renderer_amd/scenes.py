"""Deterministic synthetic scenes for the BASELINE.json configs.

The reference ships no meshes or textures (/root/reference/.gitignore:16; phong.rs:166-171 loads
assets that are not in the tree), so every config uses a procedural stand-in with the matching
triangle count.  Generators are bit-stable across machines: SplitMix64 integers, IEEE double
+,-,*,/ only, and per-ring trig tables from libm `math.sin/cos` (never NumPy's SIMD loops,
whose last bit depends on the CPU's dispatch path).  Output = float32 VSInput arrays
[ntris, 3, floats_per_vertex] in the layout of include/frr.h's vertex-shader table.
"""
import math

import numpy as np

MASK = (1 << 64) - 1
SEED_BASE = 0xF5EED000


def splitmix_u01(seed, n):
    """n values u = (z >> 40) * 2^-24 of SplitMix64 started at `seed` (float64 array, exact)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


# ---- config 1: single triangle -----------------------------------------

def single_triangle():
    """clip verts (-.5,-.5,.5,1) (.5,-.5,.5,1) (0,.5,.5,1); VS_CLIP layout [1,3,4]"""
    return np.array([[[-0.5, -0.5, 0.5, 1.0], [0.5, -0.5, 0.5, 1.0], [0.0, 0.5, 0.5, 1.0]]], np.float32)


def single_triangle_rgb():
    """same triangle with R,G,B vertex colours; VS_CLIP_COLOR layout [1,3,7]"""
    t = single_triangle()[0]
    col = np.eye(3, dtype=np.float32)
    return np.concatenate([t, col], axis=1)[None].astype(np.float32)


# ---- config 4: random clip-space triangles ----------------------------------------------------

def random_clip_triangles(n, width, height, seed=SEED_BASE + 4, spread=0.95, w_jitter=0.1):
    """n clip-space triangles, VS_CLIP layout [n,3,4].

    Per triangle (14 uniforms): w_c = 1+9u; NDC centre = spread*(2u-1) per axis (0.95); radius
    r_px = 2 * 2^floor(4u) * (1+u) (octave-uniform in [2,64), mean covered area ~60 px);
    r_ndc = 2 r_px / width; vertex k NDC = centre + r_ndc*(2u-1, 2u-1); w_k = w_c(1+0.1(2u-1));
    clip = (x w_k, y w_k, 0.5 w_k, w_k).  Some triangles cross the screen edge (clip path);
    spread > 1 and a large w_jitter (> 1: negative w) stress the quirky clipper for tests.
    """
    u = splitmix_u01(seed, 14 * n).reshape(n, 14)
    w_c = 1.0 + 9.0 * u[:, 0]
    cx = spread * (2.0 * u[:, 1] - 1.0)
    cy = spread * (2.0 * u[:, 2] - 1.0)
    octave = np.floor(4.0 * u[:, 3])
    r_px = 2.0 * np.ldexp(1.0, octave.astype(np.int64)) * (1.0 + u[:, 4])
    r_ndc = 2.0 * r_px / float(width)
    out = np.empty((n, 3, 4), np.float64)
    for k in range(3):
        x = cx + r_ndc * (2.0 * u[:, 5 + 3 * k] - 1.0)
        y = cy + r_ndc * (2.0 * u[:, 6 + 3 * k] - 1.0) * (float(width) / float(height))
        w = w_c * (1.0 + w_jitter * (2.0 * u[:, 7 + 3 * k] - 1.0))
        out[:, k, 0] = x * w
        out[:, k, 1] = y * w
        out[:, k, 2] = 0.5 * w
        out[:, k, 3] = w
    return out.astype(np.float32)


# ---- parametric meshes (configs 2, 3) -----------------------------------------------------------

def _ring_tables(n):
    ang = [2.0 * math.pi * i / n for i in range(n + 1)]
    c = np.array([math.cos(a) for a in ang])
    s = np.array([math.sin(a) for a in ang])
    c[n], s[n] = c[0], s[0]  # closed seam
    return c, s


def _grid_to_triangles(P, Nrm, UV):
    """[nu+1, nv+1, .] vertex grids -> [nu*nv*2, 3, 8] (pos3, uv2, normal3), two tris per quad."""
    nu, nv = P.shape[0] - 1, P.shape[1] - 1
    V = np.concatenate([P, UV, Nrm], axis=2)  # [., ., 8]
    a = V[:-1, :-1]
    b = V[1:, :-1]
    c = V[1:, 1:]
    d = V[:-1, 1:]
    t0 = np.stack([a, b, c], axis=2)
    t1 = np.stack([a, c, d], axis=2)
    tris = np.stack([t0, t1], axis=2).reshape(nu * nv * 2, 3, 8)
    return tris.astype(np.float32)


def torus(nu=56, nv=56, R=1.0, r=0.4):
    """config 2 "teapot-class": 56*56*2 = 6,272 triangles; VS_PHONG/VS_GOURAUD layout [n,3,8]."""
    cu, su = _ring_tables(nu)
    cv, sv = _ring_tables(nv)
    CU, CV = np.meshgrid(cu, cv, indexing="ij")
    SU, SV = np.meshgrid(su, sv, indexing="ij")
    P = np.stack([(R + r * CV) * CU, r * SV, (R + r * CV) * SU], axis=2)
    Nrm = np.stack([CV * CU, SV, CV * SU], axis=2)
    iu = np.arange(nu + 1) / float(nu)
    iv = np.arange(nv + 1) / float(nv)
    UV = np.stack(np.meshgrid(iu, iv, indexing="ij"), axis=2)
    return _grid_to_triangles(P, Nrm, UV)


def displaced_sphere(n=186, bump=0.08):
    """config 3 "bunny-class": 186*186*2 = 69,192 triangles; layout [n,3,8]."""
    cu, su = _ring_tables(n)               # longitude
    # latitude from pole to pole over n segments (half turn)
    lat = [math.pi * j / n for j in range(n + 1)]
    cl = np.array([math.cos(a) for a in lat])
    sl = np.array([math.sin(a) for a in lat])
    # displacement from low-order harmonics built out of the same tables (products only)
    CU, CL = np.meshgrid(cu, cl, indexing="ij")
    SU, SL = np.meshgrid(su, sl, indexing="ij")
    c2u = CU * CU - SU * SU
    s2u = 2.0 * SU * CU
    c3u = c2u * CU - s2u * SU
    rad = 1.0 + bump * (c3u * SL * SL * SL + 0.5 * s2u * SL * CL)
    D = np.stack([SL * CU, CL, SL * SU], axis=2)
    P = D * rad[:, :, None]
    iu = np.arange(n + 1) / float(n)
    UV = np.stack(np.meshgrid(iu, iu, indexing="ij"), axis=2)
    return _grid_to_triangles(P, D, UV)


def layered_sheets(gx=250, gy=100, layers=5):
    """config 5 "Sponza-class": layers*gx*gy*2 = 250,000 triangles of stacked textured sheets at
    different depths and scales (deep overdraw, large and small triangles, outer layers clipped)."""
    out = []
    for l in range(layers):
        scale = 0.6 * (1.45 ** l)
        z = 1.5 - 0.55 * l
        xs = (np.arange(gx + 1) / float(gx) - 0.5) * 2.0 * scale * (16.0 / 9.0)
        ys = (np.arange(gy + 1) / float(gy) - 0.5) * 2.0 * scale + 0.15 * l
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        wob = 0.05 * ((np.arange(gx + 1)[:, None] * 7 + np.arange(gy + 1)[None, :] * 13 + l * 5) % 11) / 11.0
        P = np.stack([X, Y, z + wob], axis=2)
        Nrm = np.stack([0.2 * (wob - 0.025) * 10.0, 0.3 * np.ones_like(X), -np.ones_like(X)], axis=2)
        UV = np.stack(np.meshgrid(np.arange(gx + 1) / float(gx), np.arange(gy + 1) / float(gy), indexing="ij"), axis=2)
        out.append(_grid_to_triangles(P, Nrm, UV))
    return np.concatenate(out, axis=0)


def checker_texture(size=1024, cell=32):
    """RGBA8 checker + gradient, integer arithmetic only; square (sample_2d needs height >= width)."""
    y, x = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    chk = (((x // cell) + (y // cell)) & 1).astype(np.uint32)
    r = (64 + chk * 160 + (x * 31 // size)).astype(np.uint8)
    g = (48 + (1 - chk) * 150 + (y * 57 // size)).astype(np.uint8)
    b = (32 + ((x * 3 + y * 5) * 223 // (8 * size))).astype(np.uint8)
    a = np.full_like(r, 255)
    return np.ascontiguousarray(np.stack([r, g, b, a], axis=2))


def demo_camera(width, height):
    """(eye, at, up, fovy, aspect, zn, zf): camera of SURVEY §8d config 2/3, projection as phong.rs:164."""
    return (0.0, 1.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.pi * 0.25, float(width) / float(height), 0.1, 100.0


# ---- the BASELINE.json configs as data (bench.py, tests/golden/make_frames.py, the parity tests, tools/) ------------

CONFIG_NAMES = ("cfg1", "cfg2", "cfg3", "cfg3b", "cfg4", "cfg5", "headline")


def build_config(name, reduced=False):
    """One BASELINE.json config as plain data: dict(name, W, H, mesh [n,3,NF] f32, vs, ps (names of
    the shader-table entries), cam (use demo_camera), tex (RGBA8 or None), flat_color).  `reduced`: the same
    scene at 1/8 of the linear frame size with a coarser mesh -- what the CPU-only checkers
    can afford."""
    flat = (1.0, 0.5, 0.25, 1.0)
    r = bool(reduced)
    if name == "cfg1":
        W = H = 64 if r else 512
        return dict(name=name, W=W, H=H, mesh=single_triangle(), vs="CLIP", ps="FLAT", cam=False, tex=None, flat_color=flat)
    if name == "cfg2":
        W, H = (240, 135) if r else (1920, 1080)
        return dict(name=name, W=W, H=H, mesh=torus(12, 10) if r else torus(), vs="GOURAUD", ps="COLOR", cam=True, tex=None, flat_color=flat)
    if name in ("cfg3", "cfg3b"):
        W, H = (240, 135) if r else (1920, 1080)
        tex = checker_texture(64, 8) if r else checker_texture(1024, 32)
        return dict(name=name, W=W, H=H, mesh=displaced_sphere(n=16 if r else 186), vs="PHONG", ps="PHONG" if name == "cfg3" else "BLINN",
                    cam=True, tex=tex, flat_color=flat)
    if name == "cfg4":
        W = H = 512 if r else 4096
        n = 2000 if r else 1_000_000
        return dict(name=name, W=W, H=H, mesh=random_clip_triangles(n, W, H), vs="CLIP", ps="DEPTH", cam=False, tex=None, flat_color=flat)
    if name == "headline":
        W, H = (240, 135) if r else (1920, 1080)
        n = 2000 if r else 1_000_000
        return dict(name=name, W=W, H=H, mesh=random_clip_triangles(n, W, H), vs="CLIP", ps="DEPTH", cam=False, tex=None, flat_color=flat)
    if name == "cfg5":
        W, H = (480, 270) if r else (3840, 2160)
        tex = checker_texture(64, 8) if r else checker_texture(1024, 32)
        mesh = layered_sheets(10, 6, 5) if r else layered_sheets()
        return dict(name=name, W=W, H=H, mesh=mesh, vs="PHONG", ps="BLINN", cam=True, tex=tex, flat_color=flat)
    raise KeyError(name)


CONFIGS = {
    "cfg1_single_triangle": dict(width=512, height=512),
    "cfg2_torus_gouraud": dict(width=1920, height=1080),
    "cfg3_sphere_phong": dict(width=1920, height=1080),
    "cfg4_random_1m_depth": dict(width=4096, height=4096, ntris=1_000_000),
    "headline_random_1m_1080p": dict(width=1920, height=1080, ntris=1_000_000),
    "cfg5_sheets_phong_4k": dict(width=3840, height=2160),
}
