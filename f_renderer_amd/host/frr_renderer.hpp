// frr_renderer.hpp -- header-only C++17 host mirror of the reference's rasterization interface
// over the C ABI of include/frr.h (libfrr_hip.so).  No Rust toolchain exists in this image, so this
// is the compiled-language host layer: same names, argument meaning and error behaviour as
//   Renderer::geometry_processing  /root/reference/f_renderer/src/renderer.rs:96-112
//   Renderer::rasterization        /root/reference/f_renderer/src/renderer.rs:269-284
//   FrameBuffer::{new,fill,clear,get_size,get_data,set_pixel,get_pixel}  renderer.rs:418-514
//   set_identity / set_look_at / set_perspective   matrix_util.rs:3-35,   Camera  camera.rs:4-26
//   Model::{new,faces_len,vert,uv,normal}  obj_loader.rs:7-97 (+ init_vertex_input, phong.rs:187-201)
//   FrameBuffer::load_file  renderer.rs:427-471 (BGRA storage; TGA decoded here, the reference uses the image crate)
// batched at the granularity of the reference's draw loop (examples/src/bin/phong.rs:314-387) and
// with table-selected shaders instead of closures.  What panics in the reference throws frr::Error.
// All arithmetic of the path runs on the GPU inside libfrr_hip.so; this file is plumbing.
#pragma once
#include <array>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/frr.h"

namespace frr {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("frr error " + std::to_string(c) + ": " + m), code(c) {}
};

using Mat4 = std::array<float, 16>; // column-major, as glam::Mat4::from_cols_array
struct Vec3 { float x, y, z; };

inline Mat4 set_identity() { Mat4 m; frr_set_identity(m.data()); return m; }                       // matrix_util.rs:3-8
inline Mat4 set_look_at(Vec3 eye, Vec3 at, Vec3 up)                                                // matrix_util.rs:10-22
{
    Mat4 m; const float e[3] = {eye.x, eye.y, eye.z}, a[3] = {at.x, at.y, at.z}, u[3] = {up.x, up.y, up.z};
    frr_set_look_at(e, a, u, m.data());
    return m;
}
inline Mat4 set_perspective(float fovy, float aspect, float zn, float zf)                          // matrix_util.rs:24-35
{
    Mat4 m; frr_set_perspective(fovy, aspect, zn, zf, m.data()); return m;
}

struct Camera {                                                                                    // camera.rs:4-26
    Vec3 eye, at, up;
    Mat4 mat_look_at;
    Camera(Vec3 e, Vec3 a, Vec3 u) : eye(e), at(a), up(u), mat_look_at(set_look_at(e, a, u)) {}
    const Mat4 &cal_look_at() { mat_look_at = set_look_at(eye, at, up); return mat_look_at; }
};

// Host image with the reference's FrameBuffer surface: RGBA8 row-major, offset (y*width + x)*4.
class FrameBuffer {
public:
    FrameBuffer(uint32_t width, uint32_t height) : width_(width), height_(height), buffer_((size_t)width * height * 4, 0) {}
    static FrameBuffer create(uint32_t width, uint32_t height) { return FrameBuffer(width, height); } // FrameBuffer::new
    // FrameBuffer::load_file (renderer.rs:427-471): decoded image rows top-down, Rgb8 / Rgba8 only, stored B,G,R,A
    // (alpha 255 for Rgb8); anything else panics in the reference -> frr::Error.  `pixels`: h x w x channels bytes.
    static FrameBuffer from_image(const uint8_t *pixels, uint32_t width, uint32_t height, int channels)
    {
        if (channels != 3 && channels != 4) throw Error(FRR_ERR_INVALID, "invalid color type (renderer.rs:461-463)");
        FrameBuffer fb(width, height);
        for (size_t i = 0; i < (size_t)width * height; ++i) {
            const uint8_t *c = pixels + i * channels;
            uint8_t *o = &fb.buffer_[i * 4];
            o[0] = c[2]; o[1] = c[1]; o[2] = c[0]; o[3] = channels == 4 ? c[3] : 255;             // :442-445, :454-457
        }
        return fb;
    }
    // True-colour TGA (types 2 and 10 = RLE, 24 or 32 bpp; the reference's assets are .tga, phong.rs:167-171)
    static FrameBuffer load_file(const std::string &path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Error(FRR_ERR_INVALID, "cannot open " + path + " (image::open(..).unwrap() panics)");
        const std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (d.size() < 18) throw Error(FRR_ERR_INVALID, "truncated TGA header");
        const int idlen = d[0], cmap = d[1], typ = d[2], bpp = d[16], desc = d[17];
        const uint32_t w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8);
        if (cmap != 0 || (typ != 2 && typ != 10) || (bpp != 24 && bpp != 32)) throw Error(FRR_ERR_UNSUPPORTED, "unsupported TGA (true-colour 24/32 bpp, types 2 and 10 only)");
        const size_t n = (size_t)w * h, bp = bpp / 8;
        size_t pos = 18 + idlen;
        std::vector<uint8_t> raw(n * bp);
        auto need = [&](size_t k) { if (pos + k > d.size()) throw Error(FRR_ERR_INVALID, "truncated TGA data"); };
        if (typ == 2) {
            need(n * bp);
            std::memcpy(raw.data(), &d[pos], n * bp);
        } else {
            for (size_t i = 0; i < n;) {
                need(1);
                const int hdr = d[pos++];
                const size_t cnt = (size_t)(hdr & 0x7F) + 1;
                if (i + cnt > n) throw Error(FRR_ERR_INVALID, "TGA run leaves the image");
                if (hdr & 0x80) {
                    need(bp);
                    for (size_t k = 0; k < cnt; ++k) std::memcpy(&raw[(i + k) * bp], &d[pos], bp);
                    pos += bp;
                } else {
                    need(cnt * bp);
                    std::memcpy(&raw[i * bp], &d[pos], cnt * bp);
                    pos += cnt * bp;
                }
                i += cnt;
            }
        }
        // file order is B,G,R(,A), rows bottom-up unless descriptor bit 5, columns right-to-left if bit 4
        std::vector<uint8_t> img(n * bp);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const uint32_t sy = (desc & 0x20) ? y : h - 1 - y, sx = (desc & 0x10) ? w - 1 - x : x;
                const uint8_t *s = &raw[((size_t)sy * w + sx) * bp];
                uint8_t *o = &img[((size_t)y * w + x) * bp];
                o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
                if (bp == 4) o[3] = s[3];
            }
        return from_image(img.data(), w, h, (int)bp);
    }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    const std::vector<uint8_t> &get_data() const { return buffer_; }
    std::vector<uint8_t> &get_data_mut() { return buffer_; }
    uint32_t get_size() const { return width_ * height_ * 4; }
    void clear() { std::fill(buffer_.begin(), buffer_.end(), 0); }
    void fill(const std::array<uint8_t, 4> &c) { for (size_t i = 0; i < buffer_.size(); i += 4) std::memcpy(&buffer_[i], c.data(), 4); }
    void set_pixel(uint32_t x, uint32_t y, const std::array<uint8_t, 4> &c) { std::memcpy(&buffer_.at((size_t)(y * width_ + x) * 4 + 0), c.data(), 4); }
    std::array<uint8_t, 4> get_pixel(uint32_t x, uint32_t y) const
    {
        std::array<uint8_t, 4> c;
        std::memcpy(c.data(), &buffer_.at((size_t)(y * width_ + x) * 4), 4);
        return c;
    }
private:
    uint32_t width_, height_;
    std::vector<uint8_t> buffer_;
};

struct VSInput { float pos[3]; float uv[2]; float normal[3]; };          // phong.rs:49-54 (FRR_VS_PHONG / GOURAUD)
static_assert(sizeof(VSInput) == 32, "VSInput must be 8 packed floats");

// obj_loader.rs:7-97.  Lines are split on "\n", tokens on SINGLE spaces (consecutive spaces give empty tokens, as
// in the reference), "\r" is stripped from the tokens that are parsed; only v / vn / vt / f lines are used; a face
// takes its first three a/b/c triples (1-based -> 0-based); normals are normalised when fetched (:93-96).  Every
// `.unwrap()` / index panic of the reference is an frr::Error here.  (Invalid UTF-8 is replaced in the reference,
// :28; such bytes can only sit in lines or tokens that are never parsed as numbers, so they need no handling.)
class Model {
public:
    explicit Model(const std::string &path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Error(FRR_ERR_INVALID, "cannot open " + path + " (File::open(..).unwrap() panics, obj_loader.rs:24)");
        const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        parse(text);
    }
    static Model from_text(const std::string &text) { Model m; m.parse(text); return m; }
    size_t faces_len() const { return faces_.size(); }                                                     // :79-81
    Vec3 vert(size_t i_face, size_t nth) const { return at(verts_, faces_.at(i_face)[nth][0]); }           // :83-86
    std::array<float, 2> uv(size_t i_face, size_t nth) const { return at(uv_, faces_.at(i_face)[nth][1]); } // :88-91
    Vec3 normal(size_t i_face, size_t nth) const                                                           // :93-96
    {
        const Vec3 n = at(norms_, faces_.at(i_face)[nth][2]);
        const float dot = (n.x * n.x + n.y * n.y) + n.z * n.z;        // glam Vec3::normalize = self * (1 / sqrt(dot))
        const float r = 1.0f / std::sqrt(dot);
        return Vec3{n.x * r, n.y * r, n.z * r};
    }
    // init_vertex_input (phong.rs:187-201): Vec<[VSInput;3]>
    std::vector<std::array<struct VSInput, 3>> vertex_inputs() const;

private:
    Model() = default;
    template <class T> static const T &at(const std::vector<T> &v, uint32_t i)
    {
        if (i >= v.size()) throw Error(FRR_ERR_INVALID, "OBJ index out of bounds (the reference panics)");
        return v[i];
    }
    static std::vector<std::string> split(const std::string &s, char sep)
    {
        std::vector<std::string> out;
        size_t b = 0;
        for (;;) {
            const size_t e = s.find(sep, b);
            out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
            if (e == std::string::npos) break;
            b = e + 1;
        }
        return out;
    }
    static std::string strip_cr(std::string t) { std::string o; for (char c : t) if (c != '\r') o.push_back(c); return o; }
    static const std::string &tok(const std::vector<std::string> &l, size_t i)
    {
        if (i >= l.size()) throw Error(FRR_ERR_INVALID, "OBJ line has too few tokens (index panic in the reference)");
        return l[i];
    }
    static float parse_f32(const std::string &raw)     // str::parse::<f32>() after .replace("\r", "")
    {
        const std::string t = strip_cr(raw);
        bool ok = !t.empty();
        for (char c : t) ok = ok && (std::isdigit((unsigned char)c) || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E' ||
                                     std::isalpha((unsigned char)c)); // letters: inf / nan / infinity spellings, checked below
        char *end = nullptr;
        const float v = ok ? std::strtof(t.c_str(), &end) : 0.0f;
        if (!ok || end != t.c_str() + t.size() || t.find_first_of("xXpP") != std::string::npos)
            throw Error(FRR_ERR_INVALID, "invalid f32 literal '" + t + "' (parse::<f32>().unwrap() panics)");
        return v;
    }
    static uint32_t parse_u32_minus_1(const std::string &raw)  // parse::<u32>().unwrap() - 1 (:62-64)
    {
        const std::string t = strip_cr(raw);
        size_t b = (!t.empty() && t[0] == '+') ? 1 : 0;
        bool ok = t.size() > b;
        unsigned long long v = 0;
        for (size_t i = b; i < t.size() && ok; ++i) { ok = std::isdigit((unsigned char)t[i]) != 0; v = v * 10 + (unsigned)(t[i] - '0'); if (v > 0xFFFFFFFFull) ok = false; }
        if (!ok || v == 0) throw Error(FRR_ERR_INVALID, "invalid u32 index '" + t + "' (parse / `- 1` underflow panics in the reference)");
        return (uint32_t)(v - 1);
    }
    void parse(const std::string &text)
    {
        for (const std::string &line : split(text, '\n')) {                                                // :29
            const std::vector<std::string> l = split(line, ' ');                                           // :32
            const std::string &tag = l[0];
            if (tag == "v") verts_.push_back(Vec3{parse_f32(tok(l, 1)), parse_f32(tok(l, 2)), parse_f32(tok(l, 3))});       // :37-43
            else if (tag == "vn") norms_.push_back(Vec3{parse_f32(tok(l, 1)), parse_f32(tok(l, 2)), parse_f32(tok(l, 3))}); // :44-50
            else if (tag == "vt") uv_.push_back({parse_f32(tok(l, 1)), parse_f32(tok(l, 2))});                              // :51-56
            else if (tag == "f") {                                                                                          // :57-69
                std::array<std::array<uint32_t, 3>, 3> tri;
                for (int i = 1; i < 4; ++i) {
                    const std::vector<std::string> vv = split(tok(l, i), '/');
                    tri[i - 1] = {parse_u32_minus_1(tok(vv, 0)), parse_u32_minus_1(tok(vv, 1)), parse_u32_minus_1(tok(vv, 2))};
                }
                faces_.push_back(tri);
            }
        }
    }
    std::vector<Vec3> verts_, norms_;
    std::vector<std::array<float, 2>> uv_;
    std::vector<std::array<std::array<uint32_t, 3>, 3>> faces_;
};

inline std::vector<std::array<VSInput, 3>> Model::vertex_inputs() const
{
    std::vector<std::array<VSInput, 3>> out(faces_len());
    for (size_t i = 0; i < faces_len(); ++i)
        for (size_t j = 0; j < 3; ++j) {
            const Vec3 p = vert(i, j), n = normal(i, j);
            const std::array<float, 2> t = uv(i, j);
            out[i][j] = VSInput{{p.x, p.y, p.z}, {t[0], t[1]}, {n.x, n.y, n.z}};
        }
    return out;
}

struct Mesh { int id = -1; uint64_t ntris = 0; int vs = 0; };

// Device-resident FrameBuffer + f32 depth buffer + u32 triangle-id buffer and the two halves of the
// reference's draw loop as batched calls (`Renderer {}` itself is stateless in the reference, :41).
class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, int device = 0, void *hip_stream = nullptr) : width_(width), height_(height)
    {
        int rc = frr_create(device, width, height, hip_stream, &ctx_);
        if (rc != FRR_OK) throw Error(rc, "frr_create failed (no gfx950 device, bad size or out of memory); there is no CPU fallback");
        std::memset(&uniforms, 0, sizeof uniforms);
        frr_set_identity(uniforms.model); frr_set_identity(uniforms.view); frr_set_identity(uniforms.proj);
        uniforms.light_pos[0] = 1.2f; uniforms.light_pos[1] = 1.0f; uniforms.light_pos[2] = 2.0f;          // phong.rs:129
        uniforms.light_color[0] = uniforms.light_color[1] = uniforms.light_color[2] = 1.0f;               // phong.rs:128
        uniforms.ambient_strength = 0.1f; uniforms.specular_strength = 0.5f;                              // phong.rs:131-132
        uniforms.flat_color[0] = uniforms.flat_color[1] = uniforms.flat_color[2] = uniforms.flat_color[3] = 1.0f;
    }
    ~Renderer() { if (ctx_) frr_destroy(ctx_); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    frr_uniforms uniforms; // VSUniform + PSUniform (phong.rs:26-47); call set_uniforms() after editing

    // Vec<[VSInput;3]> (phong.rs:187-205)
    Mesh upload_mesh(const std::vector<std::array<VSInput, 3>> &tris, int vs_id = FRR_VS_PHONG)
    {
        return upload_mesh_raw(reinterpret_cast<const float *>(tris.data()), tris.size(), vs_id);
    }
    Mesh upload_mesh_raw(const float *vs_inputs, uint64_t ntris, int vs_id)
    {
        Mesh m; m.ntris = ntris; m.vs = vs_id;
        check(frr_mesh_upload(ctx_, vs_inputs, ntris, vs_id, &m.id));
        return m;
    }
    void free_mesh(Mesh &m) { if (m.id >= 0) { check(frr_mesh_free(ctx_, m.id)); m.id = -1; } }
    void set_texture(int slot, const FrameBuffer &fb) { check(frr_texture_upload(ctx_, slot, fb.get_data().data(), fb.width(), fb.height())); }
    void set_uniforms() { check(frr_set_uniforms(ctx_, &uniforms)); }
    // The reference's closure shaders (renderer.rs:105,283) as HIP text compiled at run time (include/frr.h: user shaders):
    // returns the id to upload a mesh with (vs_id) and to draw it with (pixel_shader).
    int register_shader(const std::string &hip_source, int vs_input_floats, int num_varyings)
    {
        int id = -1;
        check(frr_shader_register(ctx_, hip_source.c_str(), vs_input_floats, num_varyings, &id));
        return id;
    }
    void set_user_uniforms(const std::vector<float> &v) { check(frr_set_user_uniforms(ctx_, v.data(), (int)v.size())); }   // what the closures would capture
    void set_option(const char *name, int64_t value) { check(frr_set_option(ctx_, name, value)); }   // dev / test switches, include/frr.h
    void set_partition(int rank, int world, bool blocked = false)
    {
        check(frr_set_partition(ctx_, rank, world));
        check(frr_set_partition_layout(ctx_, blocked ? 1 : 0));
    }

    // frame_buffer.fill(color); depth_buffer.fill(depth)   (phong.rs:316-317)
    void clear(const std::array<uint8_t, 4> &color = {30, 30, 30, 255}, float depth = 0.0f) { check(frr_clear(ctx_, color.data(), depth)); }
    // loop A (phong.rs:321-331)
    void geometry_processing(const Mesh &m) { check(frr_geometry(ctx_, m.id, nullptr)); }
    // loop B (phong.rs:361-381); width_range / height_range as renderer.rs:270-271
    void rasterization(std::pair<int32_t, int32_t> width_range, std::pair<int32_t, int32_t> height_range, int pixel_shader)
    {
        check(frr_raster(ctx_, pixel_shader, width_range.first, width_range.second, height_range.first, height_range.second));
    }
    void draw(const Mesh &m, int pixel_shader)
    {
        check(frr_draw(ctx_, m.id, pixel_shader, 0, (int32_t)width_, 0, (int32_t)height_));
    }
    void sync() { check(frr_sync(ctx_)); }
    // stream-side ordering against a caller's hipStream_t (nullptr: the renderer's own), no host wait: frame_fence -- the stream
    // waits for every frame issued so far; frame_wait -- the next write of the frame targets waits for what the stream holds now
    void frame_fence(void *stream = nullptr) { check(frr_frame_fence(ctx_, stream)); }
    void frame_wait(void *stream = nullptr) { check(frr_frame_wait(ctx_, stream)); }

    // image_slice.copy_from_slice(frame_buffer.get_data())   (phong.rs:386)
    void read_frame_buffer(FrameBuffer &fb)
    {
        if (fb.width() != width_ || fb.height() != height_) throw Error(FRR_ERR_INVALID, "FrameBuffer size mismatch");
        check(frr_readback(ctx_, fb.get_data_mut().data(), nullptr, nullptr));
    }
    void read_depth(std::vector<float> &depth) { depth.resize((size_t)width_ * height_); check(frr_readback(ctx_, nullptr, depth.data(), nullptr)); }
    void read_triangle_ids(std::vector<uint32_t> &ids) { ids.resize((size_t)width_ * height_); check(frr_readback(ctx_, nullptr, nullptr, ids.data())); }
    frr_stats stats() { frr_stats s; check(frr_get_stats(ctx_, &s)); return s; }
    frr_ctx *raw() { return ctx_; }

private:
    // negative status: throw; a positive one (warning; none defined at present): results were delivered, remembered in last_warning
    void check(int rc) { if (rc < FRR_OK) throw Error(rc, frr_last_error(ctx_)); last_warning = rc; }
public:
    int last_warning = FRR_OK;
private:
    frr_ctx *ctx_ = nullptr;
    uint32_t width_, height_;
};

} // namespace frr
