#!/usr/bin/env python3
"""Dev tool: copy what tools/refresh_profiles.sh left in gpurun_out/final/ into the committed profiles/ (r01_v5_*)."""
import glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
one = lambda pat: sorted(glob.glob(os.path.join(F, pat)))[-1]
shutil.copy(one("kt/*/*kernel_stats.csv"), os.path.join(P, "r01_v5_kernel_stats.csv"))
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, "r01_v5_bench.json"))
shutil.copy(os.path.join(F, "configs.json"), os.path.join(P, "r01_configs.json"))
os.makedirs(os.path.join(P, "r01_v5_pmc"), exist_ok=True)
for src, dst in (("fetch", "fetch_size"), ("write", "write_size"), ("sq1", "sq_pass1"), ("sq2", "sq_pass2")):
    shutil.copy(one(src + "/*/*counter_collection.csv"), os.path.join(P, "r01_v5_pmc", dst + "_counter_collection.csv"))
shutil.copy(os.path.join(F, "part.log"), os.path.join(P, "r01_v5_partition_times.txt"))
gc = "/tmp/gc_calib"; os.makedirs(gc, exist_ok=True)
shutil.copy(os.path.join(P, "r01_pmc", "gather_calib_counter_collection.csv"), gc)
subprocess.check_call([sys.executable, os.path.join(R, "tools", "make_pmc_traffic.py"), os.path.join(F, "fetch"), os.path.join(F, "write"),
                       "random_1M_tris_1920x1080_depth", os.path.join(P, "pmc_traffic.json"), gc])
subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_summary.py"), os.path.join(F, "sq1"), os.path.join(F, "sq2"),
                       "--json", os.path.join(P, "r01_v5_pmc", "sq_summary.json")], stdout=subprocess.DEVNULL)
b = json.load(open(os.path.join(P, "r01_v5_bench.json")))
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms; roofline", b["roofline"]["frac"], "traffic", b["roofline"]["traffic"])
