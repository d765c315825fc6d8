"""Build + ctypes binding of libfrr_hip.so (the C ABI declared in include/frr.h).

There is no CPU fallback: if the library cannot be built/loaded, or no gfx950 device is present,
the compute entry points raise.  PyTorch is not needed here; callers may hand in torch tensors'
device pointers and torch's current stream (plumbing only).
"""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("FRR_LIB") or os.path.join(_HERE, "libfrr_hip.so")  # FRR_LIB: developer override
_SRC = [os.path.join(_HERE, "csrc", f) for f in ("frr_api.hip", "frr_kernels.h", "frr_raster.h", "frr_device.h", "frr_exact.h")]
_HDR = os.path.join(_ROOT, "include", "frr.h")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # bit-exactness contract with the reference's Rust fp32 semantics: no FMA contraction,
    # no fast-math; IEEE div/sqrt and preserved denormals are hipcc defaults and stay on.
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # no SLP packing: a v_pk_* f32 instruction issues in the time of its two scalar halves on gfx950 and the register
    # shuffles that pair the operands come on top (tools/valu_rates.hip; -10 % VALU time in the shaded resolve)
    "-fno-slp-vectorize",
    # user shaders (frr_shader_register): the device headers' text is embedded (.incbin) and compiled at run time by hiprtc
    "-DFRR_CSRC_DIR=\"%s\"" % os.path.join(_HERE, "csrc"), "-L/opt/rocm/lib", "-lhiprtc", "-Wl,-rpath,/opt/rocm/lib",
]

FRR_OK, FRR_ERR_INVALID, FRR_ERR_HIP, FRR_ERR_NOMEM, FRR_ERR_UNSUPPORTED, FRR_ERR_CAPACITY = 0, -1, -2, -3, -4, -5
VS_CLIP, VS_CLIP_COLOR, VS_PHONG, VS_GOURAUD = 0, 1, 2, 3
PS_DEPTH, PS_FLAT, PS_COLOR, PS_PHONG, PS_BLINN = 0, 1, 2, 3, 4
MAX_VARYINGS = 16


class FrrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"frr error {code}: {msg}")
        self.code = code


class Uniforms(C.Structure):
    _fields_ = [
        ("model", C.c_float * 16), ("view", C.c_float * 16), ("proj", C.c_float * 16),
        ("view_pos", C.c_float * 3), ("light_pos", C.c_float * 3), ("light_color", C.c_float * 3),
        ("ambient_strength", C.c_float), ("specular_strength", C.c_float),
        ("flat_color", C.c_float * 4), ("texture_slot", C.c_int32),
    ]


class SetupVertex(C.Structure):
    _fields_ = [("spf", C.c_float * 2), ("spi", C.c_int32 * 2), ("rhw", C.c_float), ("ctx", C.c_float * MAX_VARYINGS)]


class Xfer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("peer", C.c_int32), ("offset", C.c_uint64), ("count", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [
        ("tris_in", C.c_uint64), ("tris_setup", C.c_uint64), ("bin_entries", C.c_uint64),
        ("frag_covered", C.c_uint64), ("frag_nan", C.c_uint64), ("draws", C.c_uint32), ("replays", C.c_uint32),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _SRC + [_HDR])


def build(force=False, verbose=False):
    """Compile f_renderer_amd/csrc for gfx950 into f_renderer_amd/libfrr_hip.so (in-tree)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH + ".tmp", _SRC[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def build_debug(out=None, defines=("FRR_DEBUG_COUNTERS",), verbose=False):
    """Dev builds (tools/): the same sources with extra -D switches, e.g. the funnel/phase counters of
    tools/debug_counters.py.  Never loaded unless FRR_LIB points at it."""
    out = out or os.path.join(_ROOT, "tools", "libfrr_dbg.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-o", out, _SRC[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


_lib = None

# name -> (restype, argtypes); every symbol include/frr.h declares
_P = C.POINTER
SIGNATURES = {
    "frr_abi_version": (C.c_int, []),
    "frr_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, _P(C.c_void_p)]),
    "frr_destroy": (None, [C.c_void_p]),
    "frr_last_error": (C.c_char_p, [C.c_void_p]),
    "frr_set_partition": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "frr_set_partition_layout": (C.c_int, [C.c_void_p, C.c_int]),
    "frr_owned_band_count": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "frr_owned_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "frr_partition_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "frr_exchange_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, _P(Xfer), C.c_int]),
    "frr_set_count_fragments": (C.c_int, [C.c_void_p, C.c_int]),
    "frr_bind_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frr_target_ptrs": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_void_p)]),
    "frr_mesh_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, _P(C.c_int)]),
    "frr_mesh_bind_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, _P(C.c_int)]),
    "frr_mesh_free": (C.c_int, [C.c_void_p, C.c_int]),
    "frr_texture_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]),
    "frr_set_uniforms": (C.c_int, [C.c_void_p, _P(Uniforms)]),
    "frr_shader_register": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, _P(C.c_int)]),
    "frr_set_user_uniforms": (C.c_int, [C.c_void_p, _P(C.c_float), C.c_int]),
    "frr_vs_input_floats": (C.c_int, [C.c_int]),
    "frr_vs_num_varyings": (C.c_int, [C.c_int]),
    "frr_clear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float]),
    "frr_geometry": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_uint64)]),
    "frr_raster": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frr_draw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "frr_frame_fence": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frr_frame_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frr_sync": (C.c_int, [C.c_void_p]),
    "frr_readback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frr_readback_setup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, _P(C.c_uint64)]),
    "frr_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "frr_event_record": (C.c_int, [C.c_void_p, C.c_int]),
    "frr_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _P(C.c_float)]),
    "frr_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "frr_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "frr_profile_set_period": (C.c_int, [C.c_void_p, C.c_uint32]),
    "frr_profile_reset": (C.c_int, [C.c_void_p]),
    "frr_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_float), _P(C.c_uint32)]),
    "frr_set_identity": (None, [_P(C.c_float)]),
    "frr_set_look_at": (None, [_P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "frr_set_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _P(C.c_float)]),
    "frr_debug_atan2f": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "frr_host_atan2f": (C.c_float, [C.c_float, C.c_float]),
    "frr_debug_scan64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "frr_debug_rcp_check": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "frr_debug_gather_calib": (C.c_int, [C.c_void_p, C.c_uint32]),
    "frr_debug_mvp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _P(C.c_float)]),
}


def lib():
    """Load libfrr_hip.so; raises (loudly) if it is missing and cannot be built."""
    global _lib
    if _lib is None:
        if not os.environ.get("FRR_LIB") and needs_build():
            build()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
