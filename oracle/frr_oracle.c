/*
 * frr_oracle.c -- CPU ORACLE (test infrastructure only; see frr_oracle.h header comment).
 *
 * Plain-C restatement of /root/reference/f_renderer/src/renderer.rs and the shaders / draw loop
 * of /root/reference/examples/src/bin/phong.rs.  PARITY UNPINNED (no reference fixtures, no Rust
 * toolchain): see frr_oracle.h and DESIGN.md.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Every float expression keeps the reference's association; `+`/`*` chains are left-to-right.
 */
#include "frr_oracle.h"
#include <math.h>
#include <string.h>

/* ---- Rust scalar semantics -------------------------------------------------------------- */

/* `f32 as i32`: truncate toward zero, saturate, NaN -> 0 (Rust reference, "as" casts) */
static int32_t f32_as_i32(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
/* `f32 as u32` */
static uint32_t f32_as_u32(float f)
{
    if (!(f > 0.0f)) return 0; /* NaN, negatives, zero */
    if (f >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)f;
}
/* `f32 as u8` */
static uint8_t f32_as_u8(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
/* f32::clamp(min,max): NaN stays NaN */
static float f32_clamp(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* f32::max: a NaN operand yields the other one */
static float f32_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
/* f32::total_cmp as a sortable integer key */
static int32_t total_order_key(float f)
{
    int32_t i;
    memcpy(&i, &f, 4);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
/* wrapping i32 ops (release-build Rust) */
static int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
static int32_t i32_clamp(int32_t x, int32_t lo, int32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
static int32_t i32_min(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t i32_max(int32_t a, int32_t b) { return a > b ? a : b; }

/* ---- glam pieces (A.7; glam ^0.21, f_renderer/Cargo.toml:9 -- not in /root/reference) ---- */

static float v3_dot(const float a[3], const float b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static void v3_sub(const float a[3], const float b[3], float o[3])
{
    o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
/* Vec3::normalize = self * (1.0 / sqrt(dot(self,self))) */
static void v3_normalize(const float a[3], float o[3])
{
    float r = 1.0f / sqrtf(v3_dot(a, a));
    o[0] = a[0] * r; o[1] = a[1] * r; o[2] = a[2] * r;
}
/* Vec3::cross */
static void v3_cross(const float a[3], const float b[3], float o[3])
{
    float x = a[1] * b[2] - b[1] * a[2];
    float y = a[2] * b[0] - b[2] * a[0];
    float z = a[0] * b[1] - b[0] * a[1];
    o[0] = x; o[1] = y; o[2] = z;
}

void o_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r)
        out[r] = ((m[0 + r] * v[0] + m[4 + r] * v[1]) + m[8 + r] * v[2]) + m[12 + r] * v[3];
}
void o_mat4_mul(const float a[16], const float b[16], float out[16])
{
    float t[16];
    for (int c = 0; c < 4; ++c) o_mat4_mul_vec4(a, b + 4 * c, t + 4 * c);
    memcpy(out, t, sizeof t);
}

/* matrix_util.rs:3-8 */
void o_set_identity(float m[16])
{
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
/* matrix_util.rs:10-22 */
void o_set_look_at(const float eye[3], const float at[3], const float up[3], float m[16])
{
    float d[3], z[3], c[3], x[3], y[3];
    v3_sub(at, eye, d);
    v3_normalize(d, z);
    v3_cross(up, z, c);
    v3_normalize(c, x);
    v3_cross(z, x, y);
    m[0] = x[0]; m[1] = y[0]; m[2] = z[0]; m[3] = 0.0f;
    m[4] = x[1]; m[5] = y[1]; m[6] = z[1]; m[7] = 0.0f;
    m[8] = x[2]; m[9] = y[2]; m[10] = z[2]; m[11] = 0.0f;
    m[12] = -v3_dot(eye, x); m[13] = -v3_dot(eye, y); m[14] = -v3_dot(eye, z); m[15] = 1.0f;
}
/* matrix_util.rs:24-35 */
void o_set_perspective(float fovy, float aspect, float zn, float zf, float m[16])
{
    float fax = 1.0f / tanf(fovy * 0.5f);
    memset(m, 0, 16 * sizeof(float));
    m[0] = fax / aspect;
    m[5] = fax;
    m[10] = zf / (zf - zn);
    m[14] = (-zn * zf) / (zf - zn);
    m[11] = 1.0f;
}

/* ---- FrameBuffer (renderer.rs:411-538) ---------------------------------------------------- */

/* renderer.rs:485-494 */
void o_fb_fill(o_framebuffer *fb, const uint8_t color[4])
{
    uint64_t size = (uint64_t)fb->width * fb->height * 4;
    for (uint64_t i = 0; i < size; i += 4)
        for (int k = 0; k < 4; ++k) fb->buffer[i + k] = color[k];
}
/* phong.rs:317 */
void o_depth_fill(float *depth, uint64_t n, float v)
{
    for (uint64_t i = 0; i < n; ++i) depth[i] = v;
}
/* renderer.rs:6-14 */
void o_vec4_to_u8(const float v[4], uint8_t out[4])
{
    for (int k = 0; k < 4; ++k) out[k] = f32_as_u8(f32_clamp(v[k] * 255.0f, 0.0f, 255.0f));
}
/* get_pixel renderer.rs:505-514 + u8_array_to_vec4 :16-24 */
static int fb_texel(const o_framebuffer *fb, uint32_t x, uint32_t y, float o[4])
{
    uint32_t offset = y * fb->width * 4u + x * 4u;
    uint64_t size = (uint64_t)fb->width * fb->height * 4;
    if ((uint64_t)offset + 3 >= size) return -1; /* slice index panic */
    for (int k = 0; k < 4; ++k) o[k] = (float)fb->buffer[offset + k] / 255.0f;
    return 0;
}
/* renderer.rs:516-538 */
int o_sample_2d(const o_framebuffer *fb, float u, float v, float out[4])
{
    float x = u * (float)fb->width;
    float y = v * (float)fb->height;
    float a = x - truncf(x); /* f32::fract */
    float b = y - truncf(y);
    uint32_t wm1 = fb->width - 1u;
    uint32_t x1 = f32_as_u32(x); if (x1 > wm1) x1 = wm1;
    uint32_t y1 = f32_as_u32(y); if (y1 > wm1) y1 = wm1; /* sic: width, :523 */
    uint32_t x2 = x1 + 1u; if (x2 > wm1) x2 = wm1;
    uint32_t y2 = y1 + 1u; if (y2 > wm1) y2 = wm1;       /* sic: width, :525 */
    float t11[4], t12[4], t21[4], t22[4];
    if (fb_texel(fb, x1, y1, t11) || fb_texel(fb, x1, y2, t12) || fb_texel(fb, x2, y1, t21) ||
        fb_texel(fb, x2, y2, t22))
        return -1;
    float oma = 1.0f - a, omb = 1.0f - b;
    for (int k = 0; k < 4; ++k) {
        float c11 = t11[k] * oma * omb;
        float c12 = t12[k] * oma * b;
        float c21 = t21[k] * a * omb;
        float c22 = t22[k] * a * b;
        out[k] = c11 + c12 + c21 + c22;
    }
    return 0;
}

/* ---- shader table ------------------------------------------------------------------------ */

int o_vs_input_floats(int vs_id)
{
    switch (vs_id) {
    case O_VS_CLIP: return 4;        /* clip xyzw */
    case O_VS_CLIP_COLOR: return 7;  /* clip xyzw, rgb */
    case O_VS_PHONG: return 8;       /* pos3, uv2, normal3  (phong.rs:49-54) */
    case O_VS_GOURAUD: return 8;
    }
    return -1;
}
int o_vs_num_varyings(int vs_id)
{
    switch (vs_id) {
    case O_VS_CLIP: return 0;
    case O_VS_CLIP_COLOR: return 3;
    case O_VS_PHONG: return 8;       /* uv2, normal3, pos3  (phong.rs:64-69) */
    case O_VS_GOURAUD: return 3;
    }
    return -1;
}

/* vertex shader contract renderer.rs:105,116; VS_PHONG follows phong.rs:114-126 */
static void vertex_shader(int vs_id, const o_uniforms *u, const float *in, float *ctx, float pos[4])
{
    switch (vs_id) {
    case O_VS_CLIP:
        pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = in[3];
        break;
    case O_VS_CLIP_COLOR:
        pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = in[3];
        ctx[0] = in[4]; ctx[1] = in[5]; ctx[2] = in[6];
        break;
    case O_VS_PHONG: {
        float pv[16], mvp[16], p[4] = {in[0], in[1], in[2], 1.0f}, w[4];
        o_mat4_mul(u->proj, u->view, pv);   /* proj * view * model, left-assoc (:119) */
        o_mat4_mul(pv, u->model, mvp);
        ctx[0] = in[3]; ctx[1] = in[4];                   /* uv      (:120) */
        ctx[2] = in[5]; ctx[3] = in[6]; ctx[4] = in[7];   /* normal  (:121-122) */
        o_mat4_mul_vec4(u->model, p, w);                  /* (:123) */
        ctx[5] = w[0]; ctx[6] = w[1]; ctx[7] = w[2];      /* (:124) */
        o_mat4_mul_vec4(mvp, p, pos);                     /* (:125) */
        break;
    }
    case O_VS_GOURAUD: {
        /* no reference arithmetic exists for Gouraud: per-vertex Lambert,
         * colour = ambient + max(dot(n,l),0) * light_color, built from the Phong PS pieces */
        float pv[16], mvp[16], p[4] = {in[0], in[1], in[2], 1.0f}, w[4];
        float n[3], l[3], d[3];
        o_mat4_mul(u->proj, u->view, pv);
        o_mat4_mul(pv, u->model, mvp);
        o_mat4_mul_vec4(u->model, p, w);
        v3_normalize(in + 5, n);
        v3_sub(u->light_pos, w, d);
        v3_normalize(d, l);
        float diff = f32_max(v3_dot(n, l), 0.0f);
        for (int k = 0; k < 3; ++k)
            ctx[k] = u->light_color[k] * u->ambient_strength + diff * u->light_color[k];
        o_mat4_mul_vec4(mvp, p, pos);
        break;
    }
    }
}

/* pixel shader contract renderer.rs:283,380; PS_PHONG follows phong.rs:133-154 */
int o_pixel_shader(int ps_id, const o_uniforms *u, const float *ctx, float out[4])
{
    switch (ps_id) {
    case O_PS_DEPTH:
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return 0;
    case O_PS_FLAT:
        for (int k = 0; k < 4; ++k) out[k] = u->flat_color[k];
        return 0;
    case O_PS_COLOR:
        out[0] = ctx[0]; out[1] = ctx[1]; out[2] = ctx[2]; out[3] = 1.0f;
        return 0;
    case O_PS_PHONG:
    case O_PS_BLINN: {
        const float *uv = ctx, *normal = ctx + 2, *wpos = ctx + 5;
        float ambient[3], n[3], l[3], v[3], d[3], diffuse[3], specular[3], tex[4];
        for (int k = 0; k < 3; ++k) ambient[k] = u->light_color[k] * u->ambient_strength; /* :134 */
        v3_normalize(normal, n);                                       /* :136 */
        v3_sub(u->light_pos, wpos, d); v3_normalize(d, l);             /* :137 */
        float diff = f32_max(v3_dot(n, l), 0.0f);                      /* :138 */
        for (int k = 0; k < 3; ++k) diffuse[k] = diff * u->light_color[k]; /* :139 */
        v3_sub(u->view_pos, wpos, d); v3_normalize(d, v);              /* :141 */
        float s;
        if (ps_id == O_PS_PHONG) {
            /* reflect(-light_dir, normal), vector_util.rs:5-7: (2.0 * L.dot(N) * N - L).normalize() */
            float L[3] = {-l[0], -l[1], -l[2]}, r[3], rn[3];
            float t = 2.0f * v3_dot(L, n);
            for (int k = 0; k < 3; ++k) r[k] = t * n[k] - L[k];
            v3_normalize(r, rn);
            s = f32_max(v3_dot(v, rn), 0.0f);                          /* :143 */
        } else {
            /* Blinn-Phong half vector: not in the reference; h = normalize(l + v) */
            float h[3] = {l[0] + v[0], l[1] + v[1], l[2] + v[2]}, hn[3];
            v3_normalize(h, hn);
            s = f32_max(v3_dot(n, hn), 0.0f);
        }
        /* powi(32): five successive squarings (compiler-rt __powisf2 / LLVM expansion) */
        s = s * s; s = s * s; s = s * s; s = s * s; s = s * s;
        for (int k = 0; k < 3; ++k) specular[k] = u->specular_strength * s * u->light_color[k]; /* :144 */
        if (!u->tex || o_sample_2d(u->tex, uv[0], uv[1], tex)) return -1; /* :146-151 */
        for (int k = 0; k < 3; ++k) out[k] = tex[k] * (ambient[k] + diffuse[k] + specular[k]); /* :153 */
        out[3] = tex[3] * 1.0f;
        return 0;
    }
    }
    return -1;
}

/* ---- geometry stage (renderer.rs:31-267) -------------------------------------------------- */

enum { X_LEFT, X_RIGHT, Y_UP, Y_DOWN, Z_NEAR, Z_FAR };            /* renderer.rs:31-39 */
static const int PLANE_LIST[6] = {X_LEFT, X_RIGHT, Y_UP, Y_DOWN, Z_NEAR, Z_FAR}; /* :123-131 */
static const float EPSILON = 1.0e-5f;                              /* :44 */

/* renderer.rs:46-58 */
static int insides(int plane, const o_vertex *v)
{
    float w = v->pos[3];
    switch (plane) {
    case X_LEFT: return v->pos[0] >= -w;
    case X_RIGHT: return v->pos[0] <= w;
    case Y_UP: return v->pos[1] <= w;
    case Y_DOWN: return v->pos[1] >= -w;
    case Z_FAR: return v->pos[2] <= v->pos[3];
    case Z_NEAR: return v->pos[2] >= 0.0f;
    }
    return 0;
}
/* renderer.rs:60-73 */
static float calculate_intersect_ratio(int plane, const o_vertex *a, const o_vertex *b)
{
    float a_w = a->pos[3], b_w = b->pos[3];
    switch (plane) {
    case X_LEFT: return -(a->pos[0] + a_w) / (b_w + b->pos[0] - a->pos[0] - a_w);
    case X_RIGHT: return (a_w - a->pos[0]) / (a_w - b_w - a->pos[0] + b->pos[0]);
    case Y_UP: return (a_w - a->pos[1]) / (a_w - b_w - a->pos[1] + b->pos[1]);
    case Y_DOWN: return -(a->pos[1] + a_w) / (b_w + b->pos[1] - a_w - a->pos[1]);
    case Z_FAR: return (a_w - a->pos[2]) / (a_w - b_w - a->pos[2] + b->pos[2]);
    case Z_NEAR: return a_w / (a_w - b_w);
    }
    return 0.0f;
}
/* renderer.rs:75-94 */
static void vertex_intersect(const o_vertex *a, const o_vertex *b, float ratio, int K, o_vertex *nv)
{
    memset(nv, 0, sizeof *nv);
    for (int k = 0; k < 4; ++k) nv->pos[k] = a->pos[k] + ratio * (b->pos[k] - a->pos[k]);
    for (int k = 0; k < K; ++k) nv->ctx[k] = a->ctx[k] + (b->ctx[k] - a->ctx[k]) * ratio;
}

int o_geometry_processing(uint32_t width, uint32_t height, const float *vs_inputs, int vs_id,
                          const o_uniforms *u, o_vertex out[O_MAX_OUT_TRIS][3])
{
    const int K = o_vs_num_varyings(vs_id), NF = o_vs_input_floats(vs_id);
    o_vertex vertices[3];
    memset(vertices, 0, sizeof vertices);                               /* :113 */
    for (int i = 0; i < 3; ++i) {                                       /* :115-121 */
        float pos[4];
        vertex_shader(vs_id, u, vs_inputs + i * NF, vertices[i].ctx, pos);
        if (pos[3] == 0.0f) return 0;
        memcpy(vertices[i].pos, pos, sizeof pos);
    }
    int inside_list[3][6];
    int all_insides = 1;
    for (int i = 0; i < 3; ++i) {                                       /* :138-148 */
        int v_all = 1;
        for (int j = 0; j < 6; ++j) {
            int in = insides(PLANE_LIST[j], &vertices[i]);
            inside_list[i][j] = in;
            v_all &= in;
        }
        all_insides &= v_all;
    }
    o_vertex valid[21];
    int n = 0;
    if (!all_insides) {                                                 /* :151-171 */
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 3; ++j)
                for (int p = 0; p < 6; ++p)
                    if (inside_list[i][p] != inside_list[j][p]) {
                        float ratio = calculate_intersect_ratio(PLANE_LIST[p], &vertices[i], &vertices[j]);
                        o_vertex nv;
                        vertex_intersect(&vertices[i], &vertices[j], ratio, K, &nv);
                        if (fabsf(nv.pos[3]) > EPSILON) valid[n++] = nv;
                    }
    }
    for (int i = 0; i < 3; ++i) valid[n++] = vertices[i];               /* :171 / :173 */
    if (n < 3) return 0;                                                /* :176 (unreachable) */

    float cx = 0.0f, cy = 0.0f;                                         /* :180-187 */
    for (int i = 0; i < n; ++i) { cx += valid[i].pos[0]; cy += valid[i].pos[1]; }
    float inv_n = 1.0f / (float)n;
    cx *= inv_n; cy *= inv_n;

    /* :205-218 stable sort by atan2 angle in [0, 2pi); keys are a pure function of the vertex */
    int32_t key[21];
    for (int i = 0; i < n; ++i) {
        float fx = valid[i].pos[0] - cx, fy = valid[i].pos[1] - cy;
        float at = atan2f(fy, fx);                                      /* f32::atan2 -> libm */
        if (at < 0.0f) at += 3.14159274101257324f * 2.0f;
        key[i] = total_order_key(at);
    }
    for (int i = 1; i < n; ++i) {                                       /* stable insertion sort */
        o_vertex tv = valid[i];
        int32_t tk = key[i];
        int j = i - 1;
        while (j >= 0 && key[j] > tk) { valid[j + 1] = valid[j]; key[j + 1] = key[j]; --j; }
        valid[j + 1] = tv; key[j + 1] = tk;
    }

    for (int i = 0; i < n; ++i) {                                       /* :220-235 */
        o_vertex *v = &valid[i];
        float w = v->pos[3];
        v->rhw = 1.0f / w;
        for (int k = 0; k < 4; ++k) v->pos[k] = v->pos[k] * v->rhw;
        v->spf[0] = (v->pos[0] + 1.0f) * (float)width * 0.5f;
        v->spf[1] = (1.0f - v->pos[1]) * (float)height * 0.5f;
        v->spi[0] = f32_as_i32(v->spf[0] + 0.5f);
        v->spi[1] = f32_as_i32(v->spf[1] + 0.5f);
    }
    if (n == 3) {                                                       /* :237-243 */
        out[0][0] = valid[0]; out[0][1] = valid[1]; out[0][2] = valid[2];
        return 1;
    }
    int nt = 0;                                                         /* :245-266 */
    int last = n - 1;
    while (last > 3) {
        out[nt][0] = valid[0]; out[nt][1] = valid[last - 1]; out[nt][2] = valid[last];
        ++nt; --last;
    }
    out[nt][0] = valid[0]; out[nt][1] = valid[2]; out[nt][2] = valid[3]; ++nt;
    out[nt][0] = valid[0]; out[nt][1] = valid[1]; out[nt][2] = valid[2]; ++nt;
    return nt;
}

/* ---- raster stage (renderer.rs:26-29, 269-384) -------------------------------------------- */

/* renderer.rs:26-29 */
static int is_top_left(const int32_t a[2], const int32_t b[2])
{
    return ((a[1] == b[1]) && (a[0] < b[0])) || (a[1] > b[1]);
}

int o_rasterization(int32_t wr0, int32_t wr1, int32_t hr0, int32_t hr1, const o_vertex tri[3],
                    int ps_id, int K, const o_uniforms *u, o_framebuffer *fb, float *depth_buffer,
                    uint64_t depth_len, uint32_t *tri_id_buf, uint32_t tri_id, o_counters *c)
{
    if (wr0 > wr1 || hr0 > hr1) return -1; /* i32::clamp asserts min <= max */
    int32_t min_x = i32_clamp(tri[0].spi[0], wr0, wr1), max_x = min_x;  /* :285-288 */
    int32_t min_y = i32_clamp(tri[0].spi[1], hr0, hr1), max_y = min_y;
    for (int k = 1; k < 3; ++k) {                                       /* :290-298 */
        min_x = i32_clamp(i32_min(min_x, tri[k].spi[0]), wr0, wr1);
        max_x = i32_clamp(i32_max(max_x, tri[k].spi[0]), wr0, wr1);
        min_y = i32_clamp(i32_min(min_y, tri[k].spi[1]), hr0, hr1);
        max_y = i32_clamp(i32_max(max_y, tri[k].spi[1]), hr0, hr1);
    }
    /* :300-312 orientation from the NDC positions */
    float v01x = tri[1].pos[0] - tri[0].pos[0], v01y = tri[1].pos[1] - tri[0].pos[1];
    float v02x = tri[2].pos[0] - tri[0].pos[0], v02y = tri[2].pos[1] - tri[0].pos[1];
    float normal_z = v01x * v02y - v02x * v01y;
    const o_vertex *vtx[3] = {&tri[0], &tri[1], &tri[2]};
    if (normal_z > 0.0f) { vtx[1] = &tri[2]; vtx[2] = &tri[1]; }

    const int32_t *p0 = vtx[0]->spi, *p1 = vtx[1]->spi, *p2 = vtx[2]->spi; /* :314-316 */
    int32_t b01 = is_top_left(p0, p1) ? 0 : 1;                          /* :318-320, :333-341 */
    int32_t b12 = is_top_left(p1, p2) ? 0 : 1;
    int32_t b20 = is_top_left(p2, p0) ? 0 : 1;

    uint64_t fbsize = (uint64_t)fb->width * fb->height * 4;
    for (int32_t cy = min_y; cy < max_y; ++cy) {                        /* :322 */
        uint64_t index_y = (uint64_t)(int64_t)(cy - hr0);               /* :323 */
        for (int32_t cx = min_x; cx < max_x; ++cx) {                    /* :324 */
            float pxx = (float)cx + 0.5f, pxy = (float)cy + 0.5f;       /* :325 */
            uint64_t index_x = (uint64_t)(int64_t)(cx - wr0);           /* :326 */
            if (c) c->bbox_px++;
            /* :329-331 */
            int32_t E01 = wadd(wmul(wneg(wsub(cx, p0[0])), wsub(p1[1], p0[1])), wmul(wsub(cy, p0[1]), wsub(p1[0], p0[0])));
            int32_t E12 = wadd(wmul(wneg(wsub(cx, p1[0])), wsub(p2[1], p1[1])), wmul(wsub(cy, p1[1]), wsub(p2[0], p1[0])));
            int32_t E20 = wadd(wmul(wneg(wsub(cx, p2[0])), wsub(p0[1], p2[1])), wmul(wsub(cy, p2[1]), wsub(p0[0], p2[0])));
            if (E01 < b01) continue;
            if (E12 < b12) continue;
            if (E20 < b20) continue;
            if (c) c->frag_covered++;

            float s0x = vtx[0]->spf[0] - pxx, s0y = vtx[0]->spf[1] - pxy; /* :343-345 */
            float s1x = vtx[1]->spf[0] - pxx, s1y = vtx[1]->spf[1] - pxy;
            float s2x = vtx[2]->spf[0] - pxx, s2y = vtx[2]->spf[1] - pxy;
            float a = fabsf(s1x * s2y - s1y * s2x);                     /* :347-349 */
            float b = fabsf(s2x * s0y - s2y * s0x);
            float cc = fabsf(s0x * s1y - s0y * s1x);
            float s = a + b + cc;                                       /* :351 */
            if (s == 0.0f) continue;
            a = a * (1.0f / s); b = b * (1.0f / s); cc = cc * (1.0f / s); /* :356-358 */

            float rhw = vtx[0]->rhw * a + vtx[1]->rhw * b + vtx[2]->rhw * cc; /* :360 */
            if (rhw != rhw && c) c->frag_nan++;

            uint64_t index = index_y * (uint64_t)(int64_t)wr1 + index_x; /* :362 */
            if (index >= depth_len) return -1;
            if (rhw < depth_buffer[index]) continue;                    /* :363-365 */
            depth_buffer[index] = rhw;                                  /* :366 */
            if (tri_id_buf) tri_id_buf[index] = tri_id;
            if (c) c->frag_zpass++;

            float w = 1.0f / (rhw != 0.0f ? rhw : 1.0f);                /* :368 */
            float c0 = vtx[0]->rhw * a * w;                             /* :370-372 */
            float c1 = vtx[1]->rhw * b * w;
            float c2 = vtx[2]->rhw * cc * w;
            float input[O_MAXK];
            for (int k = 0; k < K; ++k)                                 /* :374-378 */
                input[k] = vtx[0]->ctx[k] * c0 + vtx[1]->ctx[k] * c1 + vtx[2]->ctx[k] * c2;

            if (ps_id == O_PS_DEPTH) continue; /* depth-only table entry: no colour write */
            float color[4];
            uint8_t q[4];
            if (o_pixel_shader(ps_id, u, input, color)) return -1;      /* :380 */
            o_vec4_to_u8(color, q);
            /* set_pixel(index_x as u32, index_y as u32) :381, :496-503 */
            uint32_t offset = (uint32_t)index_y * fb->width * 4u + (uint32_t)index_x * 4u;
            if ((uint64_t)offset + 3 >= fbsize) return -1;
            for (int k = 0; k < 4; ++k) fb->buffer[offset + k] = q[k];
        }
    }
    return 0;
}

/* ---- draw loop (phong.rs:319-381) ---------------------------------------------------------- */

int64_t o_geometry_batch(uint32_t width, uint32_t height, const float *vs_inputs, uint64_t ntris,
                         int vs_id, const o_uniforms *u, o_vertex *setup_out, uint64_t setup_cap)
{
    const int NF = o_vs_input_floats(vs_id);
    uint64_t n = 0;
    o_vertex out[O_MAX_OUT_TRIS][3];
    for (uint64_t t = 0; t < ntris; ++t) {                              /* loop A :321-331 */
        int m = o_geometry_processing(width, height, vs_inputs + t * 3 * NF, vs_id, u, out);
        for (int k = 0; k < m; ++k) {
            if (setup_out) {
                if (n >= setup_cap) return -1;
                memcpy(setup_out + n * 3, out[k], 3 * sizeof(o_vertex));
            }
            ++n;
        }
    }
    return (int64_t)n;
}

int o_draw(uint32_t width, uint32_t height, int32_t wr0, int32_t wr1, int32_t hr0, int32_t hr1,
           const float *vs_inputs, uint64_t ntris, int vs_id, int ps_id, const o_uniforms *u,
           o_framebuffer *fb, float *depth_buffer, uint64_t depth_len, uint32_t *tri_id_buf,
           uint32_t tri_id_base, o_vertex *setup_out, uint64_t setup_cap, o_counters *c)
{
    const int NF = o_vs_input_floats(vs_id), K = o_vs_num_varyings(vs_id);
    if (NF < 0) return -1;
    o_vertex out[O_MAX_OUT_TRIS][3];
    uint64_t n = 0;
    if (c) c->tris_in += ntris;
    if (setup_out) {
        /* two passes exactly as phong.rs: all geometry into one Vec, then all raster */
        int64_t m = o_geometry_batch(width, height, vs_inputs, ntris, vs_id, u, setup_out, setup_cap);
        if (m < 0) return -1;
        for (int64_t i = 0; i < m; ++i) {                               /* loop B :361-381 */
            if (o_rasterization(wr0, wr1, hr0, hr1, setup_out + i * 3, ps_id, K, u, fb, depth_buffer,
                                depth_len, tri_id_buf, tri_id_base + (uint32_t)i, c))
                return -1;
        }
        n = (uint64_t)m;
    } else {
        /* same order of side effects (geometry is pure), without materialising the Vec */
        for (uint64_t t = 0; t < ntris; ++t) {
            int m = o_geometry_processing(width, height, vs_inputs + t * 3 * NF, vs_id, u, out);
            for (int k = 0; k < m; ++k) {
                if (o_rasterization(wr0, wr1, hr0, hr1, out[k], ps_id, K, u, fb, depth_buffer, depth_len,
                                    tri_id_buf, tri_id_base + (uint32_t)n, c))
                    return -1;
                ++n;
            }
        }
    }
    if (c) c->tris_setup += n;
    return 0;
}
