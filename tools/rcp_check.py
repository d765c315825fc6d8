import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import f_renderer_amd as fr
r = fr.Renderer(64, 64)
tot = 0
for lo, hi in ((0, 0x40000000), (0x40000000, 0x80000000), (0x80000000, 0xC0000000), (0xC0000000, 0xFFFFFFFF)):
    n, first = r.debug_rcp_check(lo, hi)
    print(hex(lo), hex(hi), n, hex(first))
    tot += n
print("total mismatches", tot)
