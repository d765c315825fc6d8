// frr_renderer.hpp -- header-only C++17 host mirror of the reference's rasterization interface
// over the C ABI of include/frr.h (libfrr_hip.so).  No Rust toolchain exists in this image, so this
// is the compiled-language host layer: same names, argument meaning and error behaviour as
//   Renderer::geometry_processing  /root/reference/f_renderer/src/renderer.rs:96-112
//   Renderer::rasterization        /root/reference/f_renderer/src/renderer.rs:269-284
//   FrameBuffer::{new,fill,clear,get_size,get_data,set_pixel,get_pixel}  renderer.rs:418-514
//   set_identity / set_look_at / set_perspective   matrix_util.rs:3-35,   Camera  camera.rs:4-26
// batched at the granularity of the reference's draw loop (examples/src/bin/phong.rs:314-387) and
// with table-selected shaders instead of closures.  What panics in the reference throws frr::Error.
// All arithmetic of the path runs on the GPU inside libfrr_hip.so; this file is plumbing.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
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
    void check(int rc) { if (rc != FRR_OK) throw Error(rc, frr_last_error(ctx_)); }
    frr_ctx *ctx_ = nullptr;
    uint32_t width_, height_;
};

} // namespace frr
