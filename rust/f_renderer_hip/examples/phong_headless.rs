//! UNVERIFIED (never compiled here): examples/src/bin/phong.rs:314-387 without the window, on the GPU path.
//! The scene stays in the host's hands exactly as in the reference: `Model::new` + `init_vertex_input`
//! (obj_loader.rs:15-97, phong.rs:187-201) produce `Vec<[VSInput;3]>`, `FrameBuffer::load_file` the BGRA textures.
use f_renderer_hip::{set_identity, set_look_at, set_perspective, Ps, Renderer, Uniforms, VSInput, Vs};

const WIDTH: u32 = 1920;
const HEIGHT: u32 = 1080;

fn main() -> Result<(), f_renderer_hip::Error> {
    // let body = f_renderer::obj_loader::Model::new("./assets/obj/qiyana/qiyanabody.obj");      // phong.rs:166
    // let diffuse = f_renderer::renderer::FrameBuffer::load_file(".../qiyanabody_diffuse.tga");  // phong.rs:167
    let body_vertices_input: Vec<[VSInput; 3]> = Vec::new(); // = init_vertex_input(&body)
    let (tex_w, tex_h, tex_bytes): (u32, u32, Vec<u8>) = (1, 1, vec![255; 4]); // = diffuse.get_data()

    let eye = [0.0, 1.0, 3.0];
    let mut u = Uniforms {
        model: set_identity(),                                                       // phong.rs:156
        view: set_look_at(eye, [0.0, 1.0, 0.0], [0.0, 1.0, 0.0]),                    // phong.rs:158-162
        proj: set_perspective(std::f32::consts::PI * 0.25, WIDTH as f32 / HEIGHT as f32, 0.1, 100.0), // phong.rs:164
        view_pos: eye,
        light_pos: [1.2, 1.0, 2.0],                                                  // phong.rs:129
        light_color: [1.0, 1.0, 1.0],                                                // phong.rs:128
        ambient_strength: 0.1,                                                       // phong.rs:131
        specular_strength: 0.5,                                                      // phong.rs:132
        flat_color: [1.0; 4],
        texture_slot: 0,                                                             // PSUniform.place
    };

    let mut gpu = Renderer::new(WIDTH, HEIGHT, 0)?;
    let body = gpu.upload_mesh(&body_vertices_input, Vs::Phong)?;
    gpu.set_texture(0, &tex_bytes, tex_w, tex_h)?;
    let mut frame = vec![0u8; (WIDTH * HEIGHT * 4) as usize];                        // FrameBuffer::new (phong.rs:207)

    // per frame (phong.rs:314-387)
    gpu.clear([30, 30, 30, 255], 0.0)?;                                              // :316-317
    u.texture_slot = 0;                                                              // place = BODY (:364-370)
    gpu.set_uniforms(&u)?;
    gpu.geometry_processing(&body)?;                                                 // loop A :321-331
    gpu.rasterization((0, WIDTH as i32), (0, HEIGHT as i32), Ps::Phong)?;            // loop B :361-381
    gpu.read_frame_buffer(&mut frame, None)?;                                        // :386
    println!("{:?}", gpu.stats()?);
    Ok(())
}
