for g in 1 2 3 5 7 13; do echo "cfg2 G=$g"; FRR_BIN_G=$g python tools/run_configs.py --only cfg2 --skip-oracle 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print(r['gpu_ms'], r['kernels_us'])"; done
for g in 23 34 68 136; do echo "cfg3 G=$g"; FRR_BIN_G=$g python tools/run_configs.py --only 'cfg3 ' --skip-oracle 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print(r['gpu_ms'], r['kernels_us'])"; done
for g in 41 82 123 245; do echo "cfg5 G=$g"; FRR_BIN_G=$g python tools/run_configs.py --only cfg5 --skip-oracle 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print(r['gpu_ms'], r['kernels_us'])"; done
