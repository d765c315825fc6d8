/*
 * frr_oracle.h -- CPU ORACLE for the f_renderer rasterization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * there only as the checker / the timed CPU baseline.  The product (f_renderer_amd/) never
 * links or imports it.
 *
 * What it is: a plain-C, single-threaded, line-by-line restatement of the algorithm in
 *   /root/reference/f_renderer/src/renderer.rs   (geometry_processing :96-267,
 *                                                 rasterization :269-384, FrameBuffer :411-538,
 *                                                 quantise :6-24, is_top_left :26-29)
 *   /root/reference/examples/src/bin/phong.rs    (VS :114-126, PS :133-154, draw loop :314-387)
 *   /root/reference/f_renderer/src/matrix_util.rs:3-35, vector_util.rs:5-7, camera.rs:11-25
 * Each function cites the lines it follows.  fp32 only, no FMA contraction, no reassociation
 * (compile with -ffp-contract=off, no -ffast-math), IEEE division and sqrt, denormals kept,
 * Rust `as` cast semantics (truncate, saturate, NaN -> 0), wrapping i32 arithmetic.
 *
 * PARITY UNPINNED: the reference is Rust (no rustc/cargo in this image, crates.io unreachable),
 * ships no golden vectors or tests for this path (examples/src/lib.rs:1-8 is `2+2==4`), and its
 * vector maths comes from the un-vendored crate glam ^0.21 (f_renderer/Cargo.toml:9).  The oracle
 * is therefore pinned only by (a) integer known-answer vectors derivable by hand from
 * renderer.rs:285-341 (tests/golden/kat_*.json), (b) agreement with an independent NumPy-fp32
 * restatement (oracle/oracle_np.py), (c) glam's published operator associations restated in
 * the comments below.  See DESIGN.md "Oracle".
 */
#ifndef FRR_ORACLE_H
#define FRR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O_MAXK 16          /* max f32 varyings ("ShaderContext" as a K-vector, renderer.rs:97-102) */
#define O_MAX_OUT_TRIS 19  /* 3 + 18 intersection vertices -> 21-gon -> 19 fan triangles */

/* shader table ids (closures cannot cross a C boundary) */
enum { O_VS_CLIP = 0, O_VS_CLIP_COLOR = 1, O_VS_PHONG = 2, O_VS_GOURAUD = 3 };
enum { O_PS_DEPTH = 0, O_PS_FLAT = 1, O_PS_COLOR = 2, O_PS_PHONG = 3, O_PS_BLINN = 4 };

/* Vertex<T>  (renderer.rs:387-394) */
typedef struct o_vertex {
    float ctx[O_MAXK]; /* context: T */
    float rhw;         /* reciprocal of w */
    float pos[4];      /* clip, then NDC after :226 */
    float spf[2];      /* screen coordinate */
    int32_t spi[2];    /* int screen coordinate */
} o_vertex;

/* FrameBuffer (renderer.rs:411-416): RGBA8 row-major; doubles as the texture type */
typedef struct o_framebuffer {
    uint32_t width, height;
    uint8_t *buffer; /* width*height*4, owned by the caller */
} o_framebuffer;

/* VSUniform (phong.rs:26-31) + PSUniform (phong.rs:41-47) + the compile-time light consts
 * (phong.rs:128-132) flattened into one POD.  Matrices are column-major as glam's
 * Mat4::from_cols_array (matrix_util.rs:5-7). */
typedef struct o_uniforms {
    float model[16], view[16], proj[16];
    float view_pos[3];
    float light_pos[3];        /* (1.2, 1.0, 2.0)  phong.rs:129 */
    float light_color[3];      /* (1, 1, 1)        phong.rs:128 */
    float ambient_strength;    /* 0.1              phong.rs:131 */
    float specular_strength;   /* 0.5              phong.rs:132 */
    float flat_color[4];       /* PS_FLAT constant */
    const o_framebuffer *tex;  /* the texture `place` selects (phong.rs:147-151) */
} o_uniforms;

typedef struct o_counters {
    uint64_t tris_in;       /* input triangles submitted */
    uint64_t tris_setup;    /* triangles after clip/fan (emission order) */
    uint64_t bbox_px;       /* pixels visited by the bbox loops (renderer.rs:322-324) */
    uint64_t frag_covered;  /* pass the three edge tests (renderer.rs:333-341) */
    uint64_t frag_zpass;    /* pass the depth test (renderer.rs:363-366) */
    uint64_t frag_nan;      /* fragments whose rhw is NaN (sticky in the reference, A.4) */
} o_counters;

/* number of f32 per input vertex / varyings K for a vertex shader id */
int o_vs_input_floats(int vs_id);
int o_vs_num_varyings(int vs_id);

/* renderer.rs:96-267.  vs_inputs: 3 vertices x o_vs_input_floats(vs_id).  Returns the number of
 * triangles written to out (0 == None). */
int o_geometry_processing(uint32_t width, uint32_t height, const float *vs_inputs, int vs_id,
                          const o_uniforms *u, o_vertex out[O_MAX_OUT_TRIS][3]);

/* renderer.rs:269-384.  tri_id_buf (optional, may be NULL) gets tri_id at every pixel this
 * triangle writes, indexed like depth_buffer.  Returns 0, or -1 if the reference would have
 * panicked (clamp with min > max, or an out-of-bounds buffer index). */
int o_rasterization(int32_t wr0, int32_t wr1, int32_t hr0, int32_t hr1, const o_vertex tri[3],
                    int ps_id, int K, const o_uniforms *u, o_framebuffer *fb, float *depth_buffer,
                    uint64_t depth_len, uint32_t *tri_id_buf, uint32_t tri_id, o_counters *c);

/* draw loop of phong.rs:319-381 for one mesh: all geometry, then all raster, in submission
 * order.  setup_out (optional) receives the concatenated setup triangles (capacity in
 * triangles); tri ids written to tri_id_buf are tri_id_base + emission index. */
int o_draw(uint32_t width, uint32_t height, int32_t wr0, int32_t wr1, int32_t hr0, int32_t hr1,
           const float *vs_inputs, uint64_t ntris, int vs_id, int ps_id, const o_uniforms *u,
           o_framebuffer *fb, float *depth_buffer, uint64_t depth_len, uint32_t *tri_id_buf,
           uint32_t tri_id_base, o_vertex *setup_out, uint64_t setup_cap, o_counters *c);

/* only the geometry half of o_draw (loop A, phong.rs:321-331): returns number of setup tris */
int64_t o_geometry_batch(uint32_t width, uint32_t height, const float *vs_inputs, uint64_t ntris,
                         int vs_id, const o_uniforms *u, o_vertex *setup_out, uint64_t setup_cap);

/* FrameBuffer::fill (renderer.rs:485-494), depth fill (phong.rs:317) */
void o_fb_fill(o_framebuffer *fb, const uint8_t color[4]);
void o_depth_fill(float *depth, uint64_t n, float v);
/* FrameBuffer::sample_2d (renderer.rs:516-538); returns -1 where the reference would index
 * out of bounds (the y clamp uses width, :523,:525) */
int o_sample_2d(const o_framebuffer *fb, float u, float v, float out[4]);
/* vec4_to_u8_array (renderer.rs:6-14) */
void o_vec4_to_u8(const float v[4], uint8_t out[4]);

/* pixel shader table entry point (for unit tests): returns 0 ok / -1 would-panic */
int o_pixel_shader(int ps_id, const o_uniforms *u, const float *ctx, float out[4]);

/* matrix_util.rs:3-35, camera.rs:11-25 */
void o_set_identity(float m[16]);
void o_set_look_at(const float eye[3], const float at[3], const float up[3], float m[16]);
void o_set_perspective(float fovy, float aspect, float zn, float zf, float m[16]);
/* glam Mat4*Mat4, Mat4*Vec4 (A.7) */
void o_mat4_mul(const float a[16], const float b[16], float out[16]);
void o_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);

#ifdef __cplusplus
}
#endif
#endif
