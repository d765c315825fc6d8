#!/usr/bin/env python3
"""Dev tool: copy what tools/refresh_profiles.sh left in gpurun_out/final4/ into the committed profiles/ (r04_*)."""
import glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final4")
P = os.path.join(R, "profiles")
TAG = "r04"
one = lambda pat: max(glob.glob(os.path.join(F, pat)), key=os.path.getmtime)   # (gpurun merges into the local directory: older refreshes may still lie there)
if not os.environ.get("PMC_ONLY"):   # (tools/refresh_pmc.sh refreshes the counter passes alone: the bench lines stay)
    shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, f"{TAG}_bench.json"))
shutil.copy(os.path.join(F, "configs.json"), os.path.join(P, f"{TAG}_configs.json"))
shutil.copy(os.path.join(F, "bench_serial.json"), os.path.join(P, f"{TAG}_bench_serial.json"))
shutil.copy(os.path.join(F, "partition_times.json"), os.path.join(P, f"{TAG}_partition_times.json"))
shutil.copy(os.path.join(F, "partition_times_serial.json"), os.path.join(P, f"{TAG}_partition_times_serial.json"))
shutil.copy(os.path.join(F, "shapes.log"), os.path.join(P, f"{TAG}_tile_kernel_shapes.txt"))
shutil.copy(os.path.join(F, "overlap_modes.txt"), os.path.join(P, f"{TAG}_overlap_modes.txt"))
shutil.copy(os.path.join(F, "frames_in_flight_kernel_timeline.txt"), os.path.join(P, f"{TAG}_frames_in_flight_kernel_timeline.txt"))
WL = {"headline": "random_1M_tris_1920x1080_depth", "cfg4": "random_1M_tris_4096x4096_depth", "cfg5": "sheets_259k_tris_3840x2160_blinn"}
os.makedirs(os.path.join(P, f"{TAG}_pmc"), exist_ok=True)
traffic = os.path.join(P, "pmc_traffic.json")
if os.path.exists(traffic):
    os.remove(traffic)
for w, name in WL.items():
    shutil.copy(one(f"kt_{w}/*/*kernel_stats.csv"), os.path.join(P, f"{TAG}_{w}_kernel_stats.csv"))
    shutil.copy(one(f"kts_{w}/*/*kernel_stats.csv"), os.path.join(P, f"{TAG}_{w}_kernel_stats_serial.csv"))
    shutil.copy(os.path.join(F, f"timeline_{w}.log"), os.path.join(P, f"{TAG}_{w}_tile_timeline.txt"))
    for src, dst in (("fetch", "fetch_size"), ("write", "write_size"), ("sq1", "sq_pass1"), ("sq2", "sq_pass2")):
        shutil.copy(one(f"{src}_{w}/*/*counter_collection.csv"), os.path.join(P, f"{TAG}_pmc", f"{w}_{dst}_counter_collection.csv"))
    import re
    m = re.search(r"'bin_entries': (\d+)", open(os.path.join(F, f"f_{w}.log")).read())   # (tools/pmc_frame.py prints the frame's statistics)
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "make_pmc_traffic.py"), os.path.join(F, f"fetch_{w}"), os.path.join(F, f"write_{w}"), name, traffic,
                           m.group(1) if m else "0", os.path.join(F, f"sq1_{w}")])
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_summary.py"), os.path.join(F, f"sq1_{w}"), os.path.join(F, f"sq2_{w}"),
                           "--json", os.path.join(P, f"{TAG}_pmc", f"{w}_sq_summary.json")], stdout=subprocess.DEVNULL)
b = json.load(open(os.path.join(P, f"{TAG}_bench.json")))
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms; roofline", b["roofline"]["frac"], "traffic", b["roofline"]["traffic"])
for s in b.get("secondary", []):
    print("  ", s["config"]["workload"], s["ms_per_step"], "ms; roofline", s.get("roofline", {}).get("frac"))
