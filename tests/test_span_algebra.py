"""The span algebra of the tile kernel, restated in NumPy (CPU): the set {dx : E(dx) >= bias} of one bbox row from the
per-edge constants the geometry kernel precomputes (frr_device.h: edge_words -- k, D, c) equals the set the reference's
integer edge functions accept (renderer.rs:329-341), for every kind of edge (A > 0, A < 0, A = 0), both biases, and
coordinates up to the span path's limit of +-8191.  Pins the algebra the device code relies on without a GPU."""
import numpy as np


def edge_words(ax, ay, bx, by, bias):
    A, B = -(by - ay), bx - ax
    D = abs(A)
    axby = A * ax + B * ay
    if A > 0:
        return -B, D, bias + axby + D - 1, 1
    if A < 0:
        return B, D, -bias - axby, 0
    return 2 * B, 0, 1 - 2 * bias - 2 * (B * ay), 0


def span_of_row(words, bx0, by0, row, bw):
    """[lo, hi) of dx in [0, bw) accepted by all three edges, the way span_edge_bound narrows it"""
    lo, hi = 0, bw
    for k, D, c, pos in words:
        m = c - D * bx0 + k * by0
        M = m + k * row
        if D == 0:                       # A = 0: all or nothing by the sign of the (odd) M
            assert M % 2 != 0
            if M < 0:
                hi = 0
            continue
        q = M // D                       # floor
        if pos:
            lo = max(lo, q)
        else:
            hi = min(hi, max(q + 1, 0))
    return lo, max(hi, lo)


def test_precomputed_edge_words_give_the_reference_coverage():
    rng = np.random.default_rng(7)
    checked = 0
    for it in range(4000):
        lim = int(rng.choice([40, 300, 8191]))
        p = rng.integers(-lim, lim + 1, size=(3, 2))
        if it % 7 == 0:
            p[1, 1] = p[0, 1]            # a horizontal edge: A = 0
        if it % 11 == 0:
            p[2, 0] = p[1, 0]            # a vertical edge: B = 0
        (p0x, p0y), (p1x, p1y), (p2x, p2y) = (int(a) for a in p[0]), (int(a) for a in p[1]), (int(a) for a in p[2])
        bias = [int(rng.integers(0, 2)) for _ in range(3)]
        words = [edge_words(p0x, p0y, p1x, p1y, bias[0]), edge_words(p1x, p1y, p2x, p2y, bias[1]), edge_words(p2x, p2y, p0x, p0y, bias[2])]
        for k, D, c, _ in words:
            assert -32768 <= k <= 32767 and 0 <= D < 32768 and abs(c) < 2 ** 30     # what the 16-bit fields and 24-bit multiplies assume
        bx0 = int(rng.integers(max(-lim, min(p0x, p1x, p2x) - 3), min(lim, max(p0x, p1x, p2x)) + 1))
        by0 = int(rng.integers(max(-lim, min(p0y, p1y, p2y) - 3), min(lim, max(p0y, p1y, p2y)) + 1))
        bw = int(rng.integers(1, 33))
        for row in range(0, int(rng.integers(1, 33))):
            cx = bx0 + np.arange(bw)
            cy = by0 + row
            e01 = -(cx - p0x) * (p1y - p0y) + (cy - p0y) * (p1x - p0x)
            e12 = -(cx - p1x) * (p2y - p1y) + (cy - p1y) * (p2x - p1x)
            e20 = -(cx - p2x) * (p0y - p2y) + (cy - p2y) * (p0x - p2x)
            cov = (e01 >= bias[0]) & (e12 >= bias[1]) & (e20 >= bias[2])          # renderer.rs:333-341
            lo, hi = span_of_row(words, bx0, by0, row, bw)
            want = np.zeros(bw, bool)
            want[lo:hi] = True
            assert np.array_equal(cov, want), (p.tolist(), bias, bx0, by0, row, bw, lo, hi)
            checked += 1
    assert checked > 20000
