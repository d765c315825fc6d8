"""Dev tool: the cases of tests/test_gpu_fuzz.py in bulk (FUZZ_N cases, FUZZ_SMALL=1: with tiny work lists so that the
library has to replay); prints every mismatch.  No retry loop: a frame is never re-issued by the caller."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
import f_renderer_amd as fr
from f_renderer_amd import scenes
from oracle import cref
from tests.test_gpu_fuzz import run_case
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2024")))
N = int(os.environ.get("FUZZ_N", "150"))
small = bool(int(os.environ.get("FUZZ_SMALL", "0")))
bad = nan_cases = replays = 0
for it in range(N):
    ok, desc, rp, has_nan = run_case(fr, scenes, cref, rng, small)
    nan_cases += int(has_nan); replays += rp
    if it % 10 == 0:
        print("case", it, "bad so far", bad, "replays", replays, flush=True)
    if not ok:
        bad += 1; print("MISMATCH", it, desc, flush=True)
print("fuzz done:", N, "cases,", nan_cases, "with NaN fragments,", replays, "replays,", bad, "bad")
