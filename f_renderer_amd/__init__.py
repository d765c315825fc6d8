"""f_renderer_amd -- MI355X (gfx950) rasterization path behind f_renderer's Renderer/FrameBuffer
surface.  Python here is plumbing over the C ABI in include/frr.h (libfrr_hip.so); the compute
path is hand-written HIP in f_renderer_amd/csrc.  No CPU fallback exists."""
from . import scenes  # noqa: F401
from ._native import (FRR_ERR_CAPACITY, FRR_ERR_HIP, FRR_ERR_INVALID, FRR_ERR_NOMEM, FRR_ERR_UNSUPPORTED,  # noqa: F401
                      FRR_OK, FrrError, build, lib)
from .renderer import (PS_BLINN, PS_COLOR, PS_DEPTH, PS_FLAT, PS_PHONG, VS_CLIP, VS_CLIP_COLOR, VS_GOURAUD,  # noqa: F401
                       VS_PHONG, Camera, FrameBuffer, Mesh, Renderer, set_identity, set_look_at, set_perspective)
