"""Pins the library's atan2f (f_renderer_amd/csrc/frr_exact.h, host build of the same source the
device runs) against this image's glibc atan2f, which is what Rust's f32::atan2 resolves to
(renderer.rs:208-209).  The exhaustive sweep (all 2^32 atanf inputs, 2e9 atan2f pairs: 0 mismatches)
was run once with the scratch program described in DESIGN.md; this test keeps a fast sample."""
import ctypes

import numpy as np


def test_host_port_matches_glibc():
    import f_renderer_amd as fr
    L = fr.lib()
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(2024)
    n = 60000
    bits_y = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    bits_x = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    y = bits_y.view(np.float32).copy()
    x = bits_x.view(np.float32).copy()
    # moderate magnitudes (the renderer's regime), ratios near the reduction breakpoints, specials
    y[: n // 2] = (rng.standard_normal(n // 2) * 3).astype(np.float32)
    x[: n // 2] = (rng.standard_normal(n // 2) * 3).astype(np.float32)
    bp = np.array([0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** 25, 2.0 ** -29], np.float32)
    for i, b in enumerate(bp):
        for d in range(-3, 4):
            k = 100 + i * 8 + d + 3
            x[k] = 1.5
            y[k] = np.nextafter(np.float32(1.5) * b, np.float32(np.inf if d > 0 else -np.inf)) if d else np.float32(1.5) * b
    spec = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-45, -1e-45, 3.4e38]
    k = 1000
    for a in spec:
        for b in spec:
            y[k], x[k] = a, b
            k += 1
    for a, b in zip(y, x):
        e = libm.atan2f(float(a), float(b))
        g = L.frr_host_atan2f(float(a), float(b))
        eb = np.float32(e).view(np.uint32)
        gb = np.float32(g).view(np.uint32)
        assert eb == gb or (np.isnan(e) and np.isnan(g)), (a, b, e, g)
