"""Runs the 64-byte-record gather calibration kernel (under rocprofv3 --pmc FETCH_SIZE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f_renderer_amd as fr
r = fr.Renderer(64, 64)
for _ in range(3):
    rc = fr.lib().frr_debug_gather_calib(r._ctx, 23)   # 8M records = 512 MiB
    assert rc == 0, rc
print("done")
