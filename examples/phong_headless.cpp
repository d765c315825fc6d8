// phong_headless.cpp -- headless equivalent of /root/reference/examples/src/bin/phong.rs:314-387
// on the C++ host mirror: clear -> per mesh { geometry_processing ; rasterization } -> get_data,
// with the window/Vulkan presentation (out of scope) replaced by a raw RGBA dump and a PPM.
//
//   phong_headless <mesh.f32> <ntris> <tex.rgba> <tex_size> <W> <H> <out.rgba> [<out.ppm>]
//       mesh.f32: ntris x 3 x 8 float32 (pos3, uv2, normal3 = VSInput, phong.rs:49-54)
//   phong_headless --assets <model.obj> <diffuse.tga> <W> <H> <out.rgba> [<out.ppm>]
//       the reference's own asset formats: Model::new (obj_loader.rs:15-97) + init_vertex_input (phong.rs:187-201)
//       and FrameBuffer::load_file (renderer.rs:427-471, BGRA storage)
//   phong_headless --dump-assets <model.obj> <diffuse.tga> <mesh_out.f32> <tex_out.rgba>
//       loaders only (no GPU): what the two loaders produce, for the CPU-side check against the Python mirror
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../f_renderer_amd/host/frr_renderer.hpp"

static std::vector<char> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::cerr << "cannot open " << path << "\n"; std::exit(2); }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "--dump-assets") {
        if (argc < 6) { std::cerr << "usage: phong_headless --dump-assets model.obj diffuse.tga mesh_out.f32 tex_out.rgba\n"; return 2; }
        try {
            const frr::Model model(argv[2]);
            const auto vin = model.vertex_inputs();
            const frr::FrameBuffer tex = frr::FrameBuffer::load_file(argv[3]);
            std::ofstream(argv[4], std::ios::binary).write(reinterpret_cast<const char *>(vin.data()), (std::streamsize)(vin.size() * sizeof(vin[0])));
            std::ofstream(argv[5], std::ios::binary).write(reinterpret_cast<const char *>(tex.get_data().data()), tex.get_size());
            std::printf("faces=%zu tex=%ux%u\n", model.faces_len(), tex.width(), tex.height());
        } catch (const frr::Error &e) { std::cerr << e.what() << "\n"; return 1; }
        return 0;
    }
    const bool assets = mode == "--assets";
    if ((assets && argc < 7) || (!assets && argc < 8)) {
        std::cerr << "usage: phong_headless mesh.f32 ntris tex.rgba tex_size W H out.rgba [out.ppm]\n"
                     "       phong_headless --assets model.obj diffuse.tga W H out.rgba [out.ppm]\n";
        return 2;
    }
    const int a0 = assets ? 4 : 5;                                             // index of W
    const uint32_t W = (uint32_t)std::atoi(argv[a0]), H = (uint32_t)std::atoi(argv[a0 + 1]);
    const char *out_rgba = argv[a0 + 2], *out_ppm = argc > a0 + 3 ? argv[a0 + 3] : nullptr;
    try {
        std::vector<std::array<frr::VSInput, 3>> vin;
        frr::FrameBuffer diffuse(1, 1);
        if (assets) {
            vin = frr::Model(argv[2]).vertex_inputs();                        // phong.rs:166, 187-201
            diffuse = frr::FrameBuffer::load_file(argv[3]);                   // phong.rs:167
        } else {
            const uint64_t ntris = std::strtoull(argv[2], nullptr, 10);
            const uint32_t ts = (uint32_t)std::atoi(argv[4]);
            const auto mesh_bytes = slurp(argv[1]);
            const auto tex_bytes = slurp(argv[3]);
            if (mesh_bytes.size() != ntris * 96 || tex_bytes.size() != (size_t)ts * ts * 4) { std::cerr << "size mismatch\n"; return 2; }
            vin.resize(ntris);
            std::memcpy(vin.data(), mesh_bytes.data(), mesh_bytes.size());
            diffuse = frr::FrameBuffer(ts, ts);                               // FrameBuffer::load_file stand-in
            std::memcpy(diffuse.get_data_mut().data(), tex_bytes.data(), tex_bytes.size());
        }

        frr::Renderer renderer(W, H);
        renderer.set_texture(0, diffuse);

        frr::Camera camera_1({0.0f, 1.0f, 3.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f});       // phong.rs:158-162 (at: SURVEY 8d)
        const frr::Mat4 proj = frr::set_perspective(3.14159274101257324f * 0.25f, (float)W / (float)H, 0.1f, 100.0f); // phong.rs:164
        const frr::Mat4 model = frr::set_identity();                                            // phong.rs:156
        std::memcpy(renderer.uniforms.model, model.data(), 64);
        std::memcpy(renderer.uniforms.view, camera_1.mat_look_at.data(), 64);
        std::memcpy(renderer.uniforms.proj, proj.data(), 64);
        renderer.uniforms.view_pos[0] = camera_1.eye.x; renderer.uniforms.view_pos[1] = camera_1.eye.y; renderer.uniforms.view_pos[2] = camera_1.eye.z;
        renderer.uniforms.texture_slot = 0;                                                     // PSUniform.place
        renderer.set_uniforms();

        frr::Mesh mesh = renderer.upload_mesh(vin, FRR_VS_PHONG);
        frr::FrameBuffer frame_buffer = frr::FrameBuffer::create(W, H);                         // phong.rs:207

        renderer.clear({30, 30, 30, 255}, 0.0f);                                                // phong.rs:316-317
        renderer.geometry_processing(mesh);                                                     // loop A
        renderer.rasterization({0, (int32_t)W}, {0, (int32_t)H}, FRR_PS_PHONG);                 // loop B
        renderer.read_frame_buffer(frame_buffer);                                               // phong.rs:386

        const frr_stats st = renderer.stats();
        std::printf("tris_in=%llu tris_setup=%llu frag_covered=%llu\n", (unsigned long long)st.tris_in,
                    (unsigned long long)st.tris_setup, (unsigned long long)st.frag_covered);
        std::ofstream(out_rgba, std::ios::binary).write(reinterpret_cast<const char *>(frame_buffer.get_data().data()), frame_buffer.get_size());
        if (out_ppm) {
            std::ofstream ppm(out_ppm, std::ios::binary);
            ppm << "P6\n" << W << " " << H << "\n255\n";
            for (uint32_t i = 0; i < W * H; ++i) ppm.write(reinterpret_cast<const char *>(&frame_buffer.get_data()[(size_t)i * 4]), 3);
        }
        // error behaviour: a range with min > max panics in the reference (i32::clamp) -> frr::Error here
        bool threw = false;
        try { renderer.rasterization({10, 5}, {0, (int32_t)H}, FRR_PS_PHONG); } catch (const frr::Error &e) { threw = e.code == FRR_ERR_INVALID; }
        if (!threw) { std::cerr << "expected FRR_ERR_INVALID\n"; return 3; }
    } catch (const frr::Error &e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
