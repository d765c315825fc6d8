"""Dev tool (GPU box): frame / tile-kernel time of several builds of the library (tools/libfrr_<tag>.so, made with
_native.build_debug(out, defines)) on the named workloads; one child process per build (FRR_LIB picks the library).
  python tools/exp_variants.py tag1,tag2,... [headline|cfg4|cfg5 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1].split(",")
names = sys.argv[2:] or ["cfg4"]
for t in tags:
    env = dict(os.environ)
    if t != "product":
        env["FRR_LIB"] = os.path.join(ROOT, "tools", f"libfrr_{t}.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_configs.py")] + names, env=env, capture_output=True, text=True)
    for ln in out.stdout.splitlines():
        print(f"{t:10s} {ln}", flush=True)
    if out.returncode:
        print(f"{t:10s} FAILED rc={out.returncode}: {out.stderr[-400:]}", flush=True)
