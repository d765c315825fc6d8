"""User shaders (frr_shader_register): the reference's closure API (renderer.rs:105,110 VS; :273,283 PS) as HIP text compiled
at run time into the library's own kernels.  A user shader that restates a built-in pair must give the built-in frame --
and so the oracle's -- bit for bit."""
import numpy as np
import pytest

from . import user_shaders

pytestmark = pytest.mark.gpu


def _phong_scene(fr, scenes, W, H):
    mesh = scenes.displaced_sphere(n=48)
    tex = scenes.checker_texture(96, 8)
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    kw = dict(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    return mesh, tex, kw, (eye, at, up, fovy, aspect, zn, zf)


@pytest.mark.parametrize("nw", [0, 16])
def test_phong_as_user_shader_equals_the_builtin_and_the_oracle(oracle, nw):
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 400, 260
    mesh, tex, kw, cam = _phong_scene(fr, scenes, W, H)
    eye, at, up, fovy, aspect, zn, zf = cam
    f = oracle.Frame(W, H)
    f.clear()
    u = oracle.make_uniforms(view=oracle.set_look_at(eye, at, up), proj=oracle.set_perspective(fovy, aspect, zn, zf),
                             view_pos=eye, tex=oracle.Texture(tex))
    f.draw(mesh, oracle.VS_PHONG, oracle.PS_PHONG, u)
    r = fr.Renderer(W, H)
    r.set_option("raster_nw", nw)
    r.set_texture(0, tex)
    r.set_uniforms(**kw)
    sid = r.register_shader(user_shaders.PHONG, 8, 8)
    assert sid >= 64
    out = {}
    for name, vs, ps in (("builtin", fr.VS_PHONG, fr.PS_PHONG), ("user", sid, sid)):
        m = r.upload_mesh(mesh, vs)
        for count in (True, False):
            r.set_count_fragments(count)
            r.clear()
            r.draw(m, ps)
            out[name, count] = r.readback()
            st = r.stats()
            assert st["tris_setup"] == f.counters.tris_setup and (not count or st["frag_covered"] == f.counters.frag_covered)
        s = r.setup_triangles()
        out[name, "setup"] = s
    for count in (True, False):
        for k in range(3):
            np.testing.assert_array_equal(out["user", count][k], out["builtin", count][k])
    np.testing.assert_array_equal(out["user", "setup"], out["builtin", "setup"])
    c, d, t = out["user", True]
    np.testing.assert_array_equal(c, f.color)
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))


@pytest.mark.parametrize("world", [1, 3, -1])
def test_vertex_colour_user_shader_on_clipped_triangles(oracle, world):
    """K = 3 varyings through the clipper (intersection vertices interpolate the user's varyings, renderer.rs:88-91) and on a
    partitioned ctx; against the oracle's VS_CLIP_COLOR / PS_COLOR frame."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    from .conftest import owned_pixel_rows
    W, H, n = 300, 200, 7000
    clip = scenes.random_clip_triangles(n, W, H, seed=71, spread=1.25, w_jitter=0.5)
    col = scenes.splitmix_u01(5, n * 9).reshape(n, 3, 3).astype(np.float32)
    tris = np.concatenate([clip, col], axis=2)
    f = oracle.Frame(W, H)
    f.clear((1, 2, 3, 4), 0.0)
    f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
    if f.counters.frag_nan:
        pytest.skip("NaN rhw")
    acc_c = np.zeros((H, W, 4), np.uint8)
    acc_t = np.zeros(W * H, np.uint32)
    sweep = world < 0                     # world -1: one rank, the brute-force tile kernel of the user's module (option raster_sweep)
    world = abs(world)
    for rank in range(world):
        r = fr.Renderer(W, H)
        if sweep:
            r.set_option("raster_sweep", 1)
        if world > 1:
            r.set_partition(rank, world, blocked=True)
        sid = r.register_shader(user_shaders.VERTEX_COLOR, 7, 3)
        r.clear((1, 2, 3, 4), 0.0)
        r.draw(r.upload_mesh(tris, sid), sid)
        c, d, t = r.readback()
        own = owned_pixel_rows(H, rank, world, True)
        acc_c[own] = c[own]
        acc_t.reshape(H, W)[own] = t.reshape(H, W)[own]
        r.close()
    np.testing.assert_array_equal(acc_t, f.tri_id)
    np.testing.assert_array_equal(acc_c, f.color)


def test_user_uniforms_stand_for_what_a_closure_captures(oracle):
    """u.user travels with each draw: two draws of one frame with different captured colours == two PS_FLAT draws."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    W, H = 256, 160
    a = scenes.random_clip_triangles(900, W, H, seed=3, spread=1.0)
    b = scenes.random_clip_triangles(700, W, H, seed=4, spread=1.0)
    f = oracle.Frame(W, H)
    f.clear()
    f.draw(a, oracle.VS_CLIP, oracle.PS_FLAT, oracle.make_uniforms(flat_color=(0.25, 0.5, 0.75, 1.0)))
    f.draw(b, oracle.VS_CLIP, oracle.PS_FLAT, oracle.make_uniforms(flat_color=(0.9, 0.1, 0.3, 0.5)), tri_id_base=int(f.counters.tris_setup))
    r = fr.Renderer(W, H)
    sid = r.register_shader(user_shaders.CAPTURED_COLOR, 4, 0)
    ma, mb = r.upload_mesh(a, sid), r.upload_mesh(b, sid)
    r.clear()
    r.set_user_uniforms([0.25, 0.5, 0.75, 1.0])
    r.draw(ma, sid)
    r.set_user_uniforms([0.9, 0.1, 0.3, 0.5])
    r.draw(mb, sid)
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, f.tri_id)
    np.testing.assert_array_equal(c, f.color)
    np.testing.assert_array_equal(d.view(np.uint32), f.depth.view(np.uint32))


def test_user_shader_errors():
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    r = fr.Renderer(64, 64)
    with pytest.raises(fr.FrrError) as e:
        r.register_shader(user_shaders.BROKEN, 4, 0)
    assert e.value.code == fr.FRR_ERR_UNSUPPORTED and "no_such_symbol" in str(e.value)      # the compiler's log reaches the caller
    sid = r.register_shader(user_shaders.CAPTURED_COLOR, 4, 0)
    m = r.upload_mesh(scenes.random_clip_triangles(10, 64, 64, seed=1), sid)
    r.clear()
    with pytest.raises(fr.FrrError) as e:
        r.draw(m, fr.PS_COLOR)                      # the built-in pixel shader wants three varyings, this vertex shader has none
    assert e.value.code == fr.FRR_ERR_INVALID
    sid3 = r.register_shader(user_shaders.VERTEX_COLOR, 7, 3)
    with pytest.raises(fr.FrrError) as e:
        r.draw(m, sid3)                             # ... and so does this user pixel shader
    assert e.value.code == fr.FRR_ERR_INVALID
    r.set_option("raster_sweep", 1)                # (the brute-force tile kernel is generated for user shaders too)
    r.draw(m, sid)
    r.set_option("raster_sweep", 0)
    r.draw(m, sid)
    r.sync()


def test_two_texture_user_pixel_shader_on_a_builtin_vertex_shader():
    """A closure over more than one texture (PSUniform holds three, phong.rs:41-47): frr::sample_2d_slot.  The mesh keeps the
    built-in VS_PHONG; the user pixel shader is held to a pixel shader written for this test on the NumPy oracle (the way the
    reference takes its closures), bit for bit -- RGBA8, depth, ids."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    from oracle import oracle_np as onp
    W, H = 200, 130
    mesh = scenes.displaced_sphere(n=20)
    texA = scenes.checker_texture(64, 8)
    texB = np.ascontiguousarray(np.roll(scenes.checker_texture(32, 4), 5, axis=1)[:, :, [2, 1, 0, 3]])
    eye, at, up, fovy, aspect, zn, zf = scenes.demo_camera(W, H)
    wa, wb = np.float32(0.625), np.float32(0.3)

    def ps(u, ctx):
        a = onp.sample_2d(texA, ctx[:, 0:2])
        b = onp.sample_2d(texB, ctx[:, [1, 0]])
        return a * wa + b * wb

    color = np.zeros((H, W, 4), np.uint8); color[:] = (30, 30, 30, 255)
    depth = np.zeros(W * H, np.float32)
    ids = np.full(W * H, 0xFFFFFFFF, np.uint32)
    u = onp.Uniforms(view=onp.set_look_at(eye, at, up), proj=onp.set_perspective(fovy, aspect, zn, zf), view_pos=eye)
    onp.draw(W, H, mesh, onp.VS_PHONG, ps, u, color, depth, ids)

    r = fr.Renderer(W, H)
    r.set_texture(0, texA)
    r.set_texture(1, texB)
    r.set_uniforms(view=fr.set_look_at(eye, at, up), proj=fr.set_perspective(fovy, aspect, zn, zf), view_pos=eye, texture_slot=0)
    sid = r.register_shader(user_shaders.TWO_TEXTURES, 8, 8)
    r.set_user_uniforms([wa, wb])
    m = r.upload_mesh(mesh, fr.VS_PHONG)
    r.clear()
    r.draw(m, sid)
    c, d, t = r.readback()
    np.testing.assert_array_equal(t, ids)
    np.testing.assert_array_equal(d.view(np.uint32), depth.view(np.uint32))
    np.testing.assert_array_equal(c, color)
    assert len(np.unique(c.reshape(-1, 4), axis=0)) > 50     # (a textured image, not a constant)
    r.close()


def test_user_vertex_shader_with_a_builtin_pixel_shader_and_saturated_coordinates(oracle):
    """The two halves of a user shader combine with the tables: a user VS (clip position + colour) drawn with the built-in
    PS_COLOR -- on the scene whose negative / tiny w gives saturated spi and wrapping edge functions (the brute-force sweep
    inside the span kernel, which is all a user-shaded triangle outside +-8191 has), on 1 and 2 ranks."""
    import f_renderer_amd as fr
    from f_renderer_amd import scenes
    from .conftest import owned_pixel_rows
    W, H, n = 320, 180, 1500
    clip = scenes.random_clip_triangles(n, W, H, seed=3, spread=1.3, w_jitter=1.5)
    col = scenes.splitmix_u01(17, n * 9).reshape(n, 3, 3).astype(np.float32)
    tris = np.concatenate([clip, col], axis=2)
    f = oracle.Frame(W, H)
    f.clear((9, 8, 7, 6), 0.0)
    f.draw(tris, oracle.VS_CLIP_COLOR, oracle.PS_COLOR, oracle.make_uniforms())
    for world in (1, 2):
        acc_c = np.zeros((H, W, 4), np.uint8)
        acc_t = np.zeros(W * H, np.uint32)
        acc_d = np.zeros(W * H, np.float32)
        for rank in range(world):
            r = fr.Renderer(W, H)
            if world > 1:
                r.set_partition(rank, world, blocked=True)
            sid = r.register_shader(user_shaders.VERTEX_COLOR, 7, 3)
            r.clear((9, 8, 7, 6), 0.0)
            r.draw(r.upload_mesh(tris, sid), fr.PS_COLOR if rank == 0 else sid)    # (both pixel shaders are the same function)
            c, d, t = r.readback()
            own = owned_pixel_rows(H, rank, world, True)
            acc_c[own] = c[own]
            acc_t.reshape(H, W)[own] = t.reshape(H, W)[own]
            acc_d.reshape(H, W)[own] = d.reshape(H, W)[own]
            r.close()
        np.testing.assert_array_equal(acc_t, f.tri_id)
        from .conftest import assert_depth_equal
        assert_depth_equal(acc_d, f.depth)
        np.testing.assert_array_equal(acc_c, f.color)
