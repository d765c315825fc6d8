"""HIP source texts of user shaders for the tests of frr_shader_register (the reference's closure API, renderer.rs:105,283,
as text that crosses the C ABI).  PHONG restates the built-in Phong pair (phong.rs:114-126, 133-154) through the public
helpers, so that a frame drawn with it must equal the built-in one bit for bit."""

PHONG = r"""
// VSInput = pos3, uv2, normal3 (phong.rs:49-54); ShaderContext = uv2, normal3, world pos3 (phong.rs:64-69)
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx)
{
    float w[4];
    ctx[0] = in[3]; ctx[1] = in[4];                          // phong.rs:120
    ctx[2] = in[5]; ctx[3] = in[6]; ctx[4] = in[7];          // phong.rs:121-122
    frr::mat4_mul_vec4(u.model, in[0], in[1], in[2], 1.0f, w);   // phong.rs:123-124
    ctx[5] = w[0]; ctx[6] = w[1]; ctx[7] = w[2];
    frr::mat4_mul_vec4(u.mvp, in[0], in[1], in[2], 1.0f, pos);   // phong.rs:125 (proj * view * model hoisted by the library)
}
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut)
{
    float amb[3];
    for (int k = 0; k < 3; ++k) amb[k] = u.light_color[k] * u.ambient_strength;      // phong.rs:134
    float nx = ctx[2], ny = ctx[3], nz = ctx[4];
    frr::normalize3(nx, ny, nz);                                                     // :136
    float lx = u.light_pos[0] - ctx[5], ly = u.light_pos[1] - ctx[6], lz = u.light_pos[2] - ctx[7];
    frr::normalize3(lx, ly, lz);                                                     // :137
    const float diff = frr::f32_max(frr::dot3(nx, ny, nz, lx, ly, lz), 0.0f);        // :138
    float vx = u.view_pos[0] - ctx[5], vy = u.view_pos[1] - ctx[6], vz = u.view_pos[2] - ctx[7];
    frr::normalize3(vx, vy, vz);                                                     // :141
    const float Lx = -lx, Ly = -ly, Lz = -lz;                                        // :142
    const float t = 2.0f * frr::dot3(Lx, Ly, Lz, nx, ny, nz);                        // vector_util.rs:6
    float rx = t * nx - Lx, ry = t * ny - Ly, rz = t * nz - Lz;
    frr::normalize3(rx, ry, rz);
    float s = frr::f32_max(frr::dot3(vx, vy, vz, rx, ry, rz), 0.0f);                 // :143
    s = s * s; s = s * s; s = s * s; s = s * s; s = s * s;                           // powi(32)
    float tex[4];
    frr::sample_2d(u, ctx[0], ctx[1], tex, u8lut);                                   // :146-151
    for (int k = 0; k < 3; ++k) {
        const float diffuse = diff * u.light_color[k];                               // :139
        const float spec = u.specular_strength * s * u.light_color[k];               // :144
        out[k] = tex[k] * (amb[k] + diffuse + spec);                                 // :153
    }
    out[3] = tex[3] * 1.0f;
}
"""

# clip position + an RGB colour per vertex, interpolated (the built-in VS_CLIP_COLOR / PS_COLOR pair)
VERTEX_COLOR = r"""
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx)
{
    pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = in[3];
    ctx[0] = in[4]; ctx[1] = in[5]; ctx[2] = in[6];
}
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut)
{
    out[0] = ctx[0]; out[1] = ctx[1]; out[2] = ctx[2]; out[3] = 1.0f;
}
"""

# what a closure would capture arrives through u.user: a constant colour (the built-in PS_FLAT with flat_color), K = 0
CAPTURED_COLOR = r"""
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx)
{
    pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = in[3];
}
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut)
{
    out[0] = u.user[0]; out[1] = u.user[1]; out[2] = u.user[2]; out[3] = u.user[3];
}
"""

BROKEN = r"""
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx) { pos[0] = no_such_symbol; }
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut) { }
"""

# a pixel shader over TWO textures (the reference's PSUniform holds three, phong.rs:41-47): slot 0 sampled at uv, slot 1
# at the swapped coordinates, blended with two captured weights; varyings = the built-in Phong vertex shader's (K = 8)
TWO_TEXTURES = r"""
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx)
{
    pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = 1.0f;     // (unused: the mesh keeps the built-in VS_PHONG)
}
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut)
{
    float a[4], b[4];
    frr::sample_2d_slot(u, 0, ctx[0], ctx[1], a, u8lut);
    frr::sample_2d_slot(u, 1, ctx[1], ctx[0], b, u8lut);
    for (int k = 0; k < 4; ++k) out[k] = a[k] * u.user[0] + b[k] * u.user[1];
}
"""
