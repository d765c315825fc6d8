"""north_star asks for MFMA on the batched 4x4 MVP x vertex-block contraction; SURVEY H2 predicts it
cannot be bit-exact (f32 MFMA = fmaf chain, glam = separately rounded products and sums).  This test
MEASURES it: the exact VALU kernel must equal the oracle's glam-order arithmetic bit for bit; the MFMA
kernel's differences are counted and reported (profiles/r01_mfma_mvp.json is written when run as a
script), which is why the parity path does not use it."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def measure():
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    from oracle import oracle_np as onp
    W, H = 1920, 1080
    mesh = scenes.displaced_sphere()                       # 69,192 triangles = 207,576 vertices
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    view, proj = fr.set_look_at(eye, at, up), fr.set_perspective(fovy, aspect, zn, zf)
    r = fr.Renderer(W, H)
    r.set_uniforms(view=view, proj=proj, view_pos=eye)
    m = r.upload_mesh(mesh, fr.VS_PHONG)
    nv = mesh.shape[0] * 3
    out = {}
    for name, flag in (("exact", 0), ("mfma", 1)):
        clip = np.empty((nv, 4), np.float32)
        ms = C.c_float()
        rc = fr.lib().frr_debug_mvp(r._ctx, m.id, flag, clip.ctypes.data, C.byref(ms))
        assert rc == 0
        out[name] = (clip, float(ms.value))
    # glam order on the host: ((c0*x + c1*y) + c2*z) + c3*w with mvp = (proj*view)*model
    mvp = onp.mat_mul(onp.mat_mul(proj, view), np.eye(4, dtype=np.float32).reshape(-1)).reshape(4, 4)
    p = mesh.reshape(-1, 8)[:, :3].astype(np.float32)
    ref = ((mvp[0][None, :] * p[:, 0:1] + mvp[1][None, :] * p[:, 1:2]) + mvp[2][None, :] * p[:, 2:3]) + mvp[3][None, :] * np.float32(1.0)
    exact, mfma = out["exact"][0], out["mfma"][0]
    ulp = np.abs(mfma.view(np.int32).astype(np.int64) - exact.view(np.int32).astype(np.int64))
    # would the integer screen position (renderer.rs:233-234) change?
    def spi(clip):
        with np.errstate(all="ignore"):
            rhw = np.float32(1.0) / clip[:, 3]
            sx = (clip[:, 0] * rhw + np.float32(1.0)) * np.float32(W) * np.float32(0.5)
            sy = (np.float32(1.0) - clip[:, 1] * rhw) * np.float32(H) * np.float32(0.5)
            return np.trunc(sx + np.float32(0.5)).astype(np.int64), np.trunc(sy + np.float32(0.5)).astype(np.int64)
    sxe, sye = spi(exact)
    sxm, sym = spi(mfma)
    return {
        "vertices": int(nv),
        "exact_equals_glam_order": bool(np.array_equal(exact.view(np.uint32), ref.astype(np.float32).view(np.uint32))),
        "mfma_components_differing": int((ulp > 0).sum()), "components": int(ulp.size),
        "mfma_vertices_differing": int((ulp > 0).any(axis=1).sum()), "mfma_max_ulp": int(ulp.max()),
        "mfma_vertices_with_different_spi": int(((sxe != sxm) | (sye != sym)).sum()),
        "kernel_ms_exact": out["exact"][1], "kernel_ms_mfma": out["mfma"][1],
    }


def test_mfma_mvp_is_not_bit_exact_but_valu_is():
    res = measure()
    print(json.dumps(res))
    assert res["exact_equals_glam_order"]
    # an fmaf chain: close (a few ulp; more only where the sum cancels towards zero), but not identical,
    # so rhw / spf would no longer be bit-exact: unusable on the parity path
    assert res["mfma_components_differing"] > 0


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = measure()
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
