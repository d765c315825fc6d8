#!/usr/bin/env python3
"""Dev tool (CPU): mean issue cost of a VALU instruction of the tile kernel, from the static instruction mix of its ISA and
the measured per-kind costs of profiles/r03_valu_issue_rates.txt (W = 4 waves per SIMD): 2.4 cycles for the fp32 add / sub /
mul / fma / fmac, integer add / sub, and / or / xor, right shifts, v_mov and v_bitop3 -- unless one source is an SGPR, then 4.2
like everything else -- and 8.2 for v_rcp / v_rsq / v_sqrt.  (A v_cndmask reading VCC right after another VCC select costs
more: not modelled, so the figure is a lower bound of the cost.)
  python tools/valu_mix.py [mangled-name substring ...]   ->  JSON {kernel: {valu, fast, slow, trans, cycles_per_valu}}"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32"}
C_FAST, C_SLOW, C_TRANS = 2.4, 4.2, 8.2


def isa():
    out = os.path.join(tempfile.gettempdir(), "frr_valu_mix.s")
    src = os.path.join(ROOT, "f_renderer_amd", "csrc", "frr_api.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-Wno-unused-function", "-DFRR_CSRC_DIR=\"%s\"" % os.path.dirname(src), "--cuda-device-only", "-S", "-o", out, src]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def mix(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    n = {"fast": 0, "slow": 0, "trans": 0}
    for l in lines[start:end]:
        t = l.strip()
        if not t.startswith("v_"):
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t.split()[0])
        ops = t.split(None, 1)[1] if " " in t else ""
        srcs = ops.split(",")[1:]
        sgpr = any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec)", x) for x in srcs)
        if op in TRANS:
            n["trans"] += 1
        elif op in FAST and not sgpr and not t.split()[0].endswith(("_dpp", "_sdwa")):
            n["fast"] += 1
        else:
            n["slow"] += 1
    v = sum(n.values())
    return dict(n, valu=v, cycles_per_valu=round((n["fast"] * C_FAST + n["slow"] * C_SLOW + n["trans"] * C_TRANS) / v, 3))


if __name__ == "__main__":
    keys = sys.argv[1:] or ["k_raster_spanILi0ELi0ELb0ELi4ELi6EE", "k_raster_spanILi0ELi0ELb0ELi3ELi6EE", "k_raster_spanILi8ELi4ELb0ELi4ELi6EE"]
    L = isa()
    print(json.dumps({k: mix(L, k) for k in keys}, indent=1))
