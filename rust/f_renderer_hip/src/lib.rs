//! UNVERIFIED: written against include/frr.h without a Rust toolchain (none exists in the build image).
//!
//! `f_renderer_hip` re-exposes the reference's rasterization surface
//!   `Renderer::geometry_processing`  f_renderer/src/renderer.rs:96-112
//!   `Renderer::rasterization`        f_renderer/src/renderer.rs:269-284
//!   `FrameBuffer::{new, fill, clear, get_data, ...}` renderer.rs:418-538
//! over the C ABI of libfrr_hip.so, batched at the granularity of the reference's own draw loop
//! (examples/src/bin/phong.rs:314-387): closures become shader-table ids, the generic `ShaderContext` a flat
//! `[f32; K]`.  The Rust host keeps owning the scene, the uniforms and the final `FrameBuffer` bytes.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub mod ffi {
    use super::*;

    #[repr(C)]
    pub struct frr_ctx {
        _private: [u8; 0],
    }

    /// include/frr.h: frr_uniforms = VSUniform (phong.rs:26-31) + PSUniform (phong.rs:41-47) + light constants
    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct frr_uniforms {
        pub model: [f32; 16], // glam::Mat4::to_cols_array()
        pub view: [f32; 16],
        pub proj: [f32; 16],
        pub view_pos: [f32; 3],
        pub light_pos: [f32; 3],
        pub light_color: [f32; 3],
        pub ambient_strength: f32,
        pub specular_strength: f32,
        pub flat_color: [f32; 4],
        pub texture_slot: i32, // PSUniform.place
    }

    #[repr(C)]
    #[derive(Clone, Copy, Default, Debug)]
    pub struct frr_stats {
        pub tris_in: u64,
        pub tris_setup: u64,
        pub bin_entries: u64,
        pub frag_covered: u64,
        pub frag_nan: u64,
        pub draws: u32,
        pub replays: u32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Default, Debug)]
    pub struct frr_xfer {
        pub kind: i32,
        pub peer: i32,
        pub offset: u64,
        pub count: u64,
    }

    extern "C" {
        pub fn frr_abi_version() -> c_int;
        pub fn frr_create(device: c_int, width: u32, height: u32, stream: *mut c_void, out: *mut *mut frr_ctx) -> c_int;
        pub fn frr_destroy(ctx: *mut frr_ctx);
        pub fn frr_last_error(ctx: *const frr_ctx) -> *const c_char;
        pub fn frr_set_option(ctx: *mut frr_ctx, name: *const c_char, value: i64) -> c_int;
        pub fn frr_set_partition(ctx: *mut frr_ctx, rank: c_int, world: c_int) -> c_int;
        pub fn frr_set_partition_layout(ctx: *mut frr_ctx, blocked: c_int) -> c_int;
        pub fn frr_owned_band_count(ctx: *const frr_ctx, y0: i32, y1: i32) -> c_int;
        pub fn frr_owned_rows(ctx: *const frr_ctx, y0: i32, y1: i32, band: i32, row0: *mut i32, row1: *mut i32) -> c_int;
        pub fn frr_bind_targets(ctx: *mut frr_ctx, color: *mut c_void, depth: *mut c_void, tri_id: *mut c_void) -> c_int;
        pub fn frr_target_ptrs(ctx: *mut frr_ctx, color: *mut *mut c_void, depth: *mut *mut c_void, tri_id: *mut *mut c_void) -> c_int;
        pub fn frr_mesh_upload(ctx: *mut frr_ctx, vs_inputs: *const f32, ntris: u64, vs_id: c_int, mesh_out: *mut c_int) -> c_int;
        pub fn frr_mesh_free(ctx: *mut frr_ctx, mesh: c_int) -> c_int;
        pub fn frr_texture_upload(ctx: *mut frr_ctx, slot: c_int, rgba: *const u8, width: u32, height: u32) -> c_int;
        pub fn frr_set_uniforms(ctx: *mut frr_ctx, u: *const frr_uniforms) -> c_int;
        pub fn frr_shader_register(ctx: *mut frr_ctx, hip_source: *const c_char, vs_input_floats: c_int, num_varyings: c_int, shader_id: *mut c_int) -> c_int;
        pub fn frr_set_user_uniforms(ctx: *mut frr_ctx, values: *const f32, n: c_int) -> c_int;
        pub fn frr_frame_fence(ctx: *mut frr_ctx, stream: *mut c_void) -> c_int;
        pub fn frr_frame_wait(ctx: *mut frr_ctx, stream: *mut c_void) -> c_int;
        pub fn frr_exchange_plan(y0: i32, y1: i32, row_elems: u32, rank: c_int, world: c_int, blocked: c_int, root: c_int, ops: *mut frr_xfer, cap: c_int) -> c_int;
        pub fn frr_partition_rows(y0: i32, y1: i32, rank: c_int, world: c_int, blocked: c_int, band: i32, row0: *mut i32, row1: *mut i32) -> c_int;
        pub fn frr_clear(ctx: *mut frr_ctx, rgba: *const u8, depth: f32) -> c_int;
        pub fn frr_geometry(ctx: *mut frr_ctx, mesh: c_int, ntris_setup: *mut u64) -> c_int;
        pub fn frr_raster(ctx: *mut frr_ctx, ps_id: c_int, x0: i32, x1: i32, y0: i32, y1: i32) -> c_int;
        pub fn frr_draw(ctx: *mut frr_ctx, mesh: c_int, ps_id: c_int, x0: i32, x1: i32, y0: i32, y1: i32) -> c_int;
        pub fn frr_sync(ctx: *mut frr_ctx) -> c_int;
        pub fn frr_readback(ctx: *mut frr_ctx, rgba: *mut u8, depth: *mut f32, tri_id: *mut u32) -> c_int;
        pub fn frr_get_stats(ctx: *mut frr_ctx, out: *mut frr_stats) -> c_int;
        pub fn frr_set_identity(m: *mut f32);
        pub fn frr_set_look_at(eye: *const f32, at: *const f32, up: *const f32, m: *mut f32);
        pub fn frr_set_perspective(fovy: f32, aspect: f32, zn: f32, zf: f32, m: *mut f32);
    }
}

pub use ffi::{frr_stats as Stats, frr_uniforms as Uniforms};

/// include/frr.h: frr_status
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
pub const ERR_INVALID: i32 = -1;
pub const ERR_HIP: i32 = -2;
pub const ERR_NOMEM: i32 = -3;
pub const ERR_UNSUPPORTED: i32 = -4;
/// never seen by a caller in practice: a work list that is too small is grown and the draw replayed inside the
/// library (renderer.rs:269-384 cannot fail); returned only after eight replays in a row were not enough
pub const ERR_CAPACITY: i32 = -5;

/// vertex-shader table: replaces the `vertex_shader: &F` closure (renderer.rs:105,110)
#[repr(i32)]
#[derive(Clone, Copy, Debug)]
pub enum Vs {
    Clip = 0,
    ClipColor = 1,
    /// phong.rs:114-126
    Phong = 2,
    Gouraud = 3,
}
/// pixel-shader table: replaces the `pixel_shader: &F` closure (renderer.rs:273,283)
#[repr(i32)]
#[derive(Clone, Copy, Debug)]
pub enum Ps {
    Depth = 0,
    Flat = 1,
    Color = 2,
    /// phong.rs:133-154
    Phong = 3,
    Blinn = 4,
}

/// `VSInput` of phong.rs:49-54 -- the FRR_VS_PHONG / FRR_VS_GOURAUD vertex layout (8 packed floats)
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct VSInput {
    pub pos: [f32; 3],
    pub uv: [f32; 2],
    pub normal: [f32; 3],
}

pub struct Mesh {
    id: c_int,
    pub ntris: u64,
}

/// Device-resident FrameBuffer + f32 depth buffer (+ u32 triangle ids) and the two halves of the reference's draw
/// loop.  The reference's `Renderer {}` is stateless (renderer.rs:41); the state here is the device mirror.
pub struct Renderer {
    ctx: *mut ffi::frr_ctx,
    pub width: u32,
    pub height: u32,
}

impl Renderer {
    pub fn new(width: u32, height: u32, device: i32) -> Result<Self, Error> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ffi::frr_create(device, width, height, std::ptr::null_mut(), &mut ctx) };
        if rc != 0 {
            return Err(Error { code: rc, message: "frr_create failed (no gfx950 device, bad size or out of memory); there is no CPU fallback".into() });
        }
        Ok(Self { ctx, width, height })
    }

    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == 0 {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(ffi::frr_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message })
    }

    /// `Vec<[VSInput;3]>` (phong.rs:187-205) -> device
    pub fn upload_mesh(&mut self, tris: &[[VSInput; 3]], vs: Vs) -> Result<Mesh, Error> {
        let mut id = -1;
        self.check(unsafe { ffi::frr_mesh_upload(self.ctx, tris.as_ptr() as *const f32, tris.len() as u64, vs as c_int, &mut id) })?;
        Ok(Mesh { id, ntris: tris.len() as u64 })
    }
    /// clip-space inputs (4 floats per vertex, `Vs::Clip`) or any other table layout, as raw floats
    pub fn upload_mesh_raw(&mut self, floats: &[f32], ntris: u64, vs: Vs) -> Result<Mesh, Error> {
        let mut id = -1;
        self.check(unsafe { ffi::frr_mesh_upload(self.ctx, floats.as_ptr(), ntris, vs as c_int, &mut id) })?;
        Ok(Mesh { id, ntris })
    }
    pub fn free_mesh(&mut self, mesh: Mesh) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_mesh_free(self.ctx, mesh.id) })
    }
    /// `PSUniform.sample_2d_*` (phong.rs:43-45): RGBA8 bytes as `FrameBuffer::get_data()` holds them (BGRA after load_file)
    pub fn set_texture(&mut self, slot: i32, rgba: &[u8], width: u32, height: u32) -> Result<(), Error> {
        assert!(rgba.len() >= (width as usize) * (height as usize) * 4);
        self.check(unsafe { ffi::frr_texture_upload(self.ctx, slot, rgba.as_ptr(), width, height) })
    }
    pub fn set_uniforms(&mut self, u: &Uniforms) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_set_uniforms(self.ctx, u) })
    }
    /// `frame_buffer.fill(color); depth_buffer.fill(depth)` (phong.rs:316-317)
    pub fn clear(&mut self, color: [u8; 4], depth: f32) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_clear(self.ctx, color.as_ptr(), depth) })
    }
    /// loop A (phong.rs:321-331): `Renderer::geometry_processing` over every triangle of `mesh`
    pub fn geometry_processing(&mut self, mesh: &Mesh) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_geometry(self.ctx, mesh.id, std::ptr::null_mut()) })
    }
    /// loop B (phong.rs:361-381): `Renderer::rasterization` with `width_range`, `height_range` (renderer.rs:270-271)
    pub fn rasterization(&mut self, width_range: (i32, i32), height_range: (i32, i32), ps: Ps) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_raster(self.ctx, ps as c_int, width_range.0, width_range.1, height_range.0, height_range.1) })
    }
    pub fn draw(&mut self, mesh: &Mesh, ps: Ps) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_draw(self.ctx, mesh.id, ps as c_int, 0, self.width as i32, 0, self.height as i32) })
    }
    /// `image_slice.copy_from_slice(frame_buffer.get_data())` (phong.rs:386)
    pub fn read_frame_buffer(&mut self, rgba: &mut [u8], depth: Option<&mut [f32]>) -> Result<(), Error> {
        let n = (self.width as usize) * (self.height as usize);
        assert!(rgba.len() >= n * 4);
        let d = match depth {
            Some(d) => {
                assert!(d.len() >= n);
                d.as_mut_ptr()
            }
            None => std::ptr::null_mut(),
        };
        self.check(unsafe { ffi::frr_readback(self.ctx, rgba.as_mut_ptr(), d, std::ptr::null_mut()) })
    }
    pub fn stats(&mut self) -> Result<Stats, Error> {
        let mut s = Stats::default();
        self.check(unsafe { ffi::frr_get_stats(self.ctx, &mut s) })?;
        Ok(s)
    }

    // ---- multi-GPU: one process per GPU, screen tile rows split over the ranks (renderer.rs:270-271's sub-window) ----
    /// Development / test switches (`include/frr.h`: `frr_set_option`); none changes a result.
    pub fn set_option(&mut self, name: &str, value: i64) -> Result<(), Error> {
        let n = std::ffi::CString::new(name).map_err(|_| Error { code: -1, message: "option name contains NUL".into() })?;
        self.check(unsafe { ffi::frr_set_option(self.ctx, n.as_ptr(), value) })
    }
    pub fn set_partition(&mut self, rank: i32, world: i32, blocked: bool) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_set_partition(self.ctx, rank, world) })?;
        self.check(unsafe { ffi::frr_set_partition_layout(self.ctx, blocked as c_int) })
    }
    /// bands `[row0, row1)` of pixel rows this rank owns: what its final-image exchange sends
    pub fn owned_rows(&self) -> Result<Vec<(i32, i32)>, Error> {
        let n = unsafe { ffi::frr_owned_band_count(self.ctx, 0, self.height as i32) };
        if n < 0 {
            return Err(Error { code: n, message: "frr_owned_band_count".into() });
        }
        let mut out = Vec::with_capacity(n as usize);
        for b in 0..n {
            let (mut r0, mut r1) = (0i32, 0i32);
            self.check(unsafe { ffi::frr_owned_rows(self.ctx, 0, self.height as i32, b, &mut r0, &mut r1) })?;
            out.push((r0, r1));
        }
        Ok(out)
    }
    /// device pointers of the render targets (RGBA8, f32 depth, u32 ids), for RCCL / peer copies
    pub fn target_ptrs(&mut self) -> Result<(*mut c_void, *mut c_void, *mut c_void), Error> {
        let (mut c, mut d, mut t) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
        self.check(unsafe { ffi::frr_target_ptrs(self.ctx, &mut c, &mut d, &mut t) })?;
        Ok((c, d, t))
    }
    /// `stream` (a hipStream_t; null = the renderer's) waits for every frame issued so far: what is enqueued on it next
    /// sees their targets (no host wait)
    pub fn frame_fence(&mut self, stream: *mut c_void) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_frame_fence(self.ctx, stream) })
    }
    /// the next write of the frame targets waits for what `stream` holds now (a reader of a target set that is bound again)
    pub fn frame_wait(&mut self, stream: *mut c_void) -> Result<(), Error> {
        self.check(unsafe { ffi::frr_frame_wait(self.ctx, stream) })
    }
}

/// The final-image exchange of one row-major plane for `rank` of `world` (blocked layout), as the list of operations one
/// RCCL group posts per frame (`examples/gather_rccl.cpp`); a pure function, no device
pub fn exchange_plan(height: i32, row_elems: u32, rank: i32, world: i32, root: i32) -> Vec<ffi::frr_xfer> {
    let n = unsafe { ffi::frr_exchange_plan(0, height, row_elems, rank, world, 1, root, std::ptr::null_mut(), 0) };
    let mut ops = vec![ffi::frr_xfer::default(); n.max(0) as usize];
    if n > 0 {
        unsafe { ffi::frr_exchange_plan(0, height, row_elems, rank, world, 1, root, ops.as_mut_ptr(), n) };
    }
    ops
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { ffi::frr_destroy(self.ctx) }
    }
}

/// matrix_util.rs:3-35, as column-major arrays (`glam::Mat4::from_cols_array`)
pub fn set_look_at(eye: [f32; 3], at: [f32; 3], up: [f32; 3]) -> [f32; 16] {
    let mut m = [0f32; 16];
    unsafe { ffi::frr_set_look_at(eye.as_ptr(), at.as_ptr(), up.as_ptr(), m.as_mut_ptr()) };
    m
}
pub fn set_perspective(fovy: f32, aspect: f32, zn: f32, zf: f32) -> [f32; 16] {
    let mut m = [0f32; 16];
    unsafe { ffi::frr_set_perspective(fovy, aspect, zn, zf, m.as_mut_ptr()) };
    m
}
pub fn set_identity() -> [f32; 16] {
    let mut m = [0f32; 16];
    unsafe { ffi::frr_set_identity(m.as_mut_ptr()) };
    m
}
