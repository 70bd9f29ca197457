"""HBM bytes of one extension from the two PMC summaries of tools/pmc_lde.py (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes; FETCH x 2 on
gfx950): python tools/lde_traffic.py <label> <fetch.json> <write.json>  -> one JSON object (measurement tool)"""
import json, sys
label, f, w = sys.argv[1], json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
logn, cols, reps = 24, 8, 3
N = 1 << logn
per, total = {}, 0.0
for k in f:
    if k not in w or not ("ntt_pass2" in k or "lde_seam" in k):
        continue
    n = f[k]["FETCH_SIZE"]["launches"] / reps
    fb, wb = 2 * f[k]["FETCH_SIZE"]["mean"] * 1024, w[k]["WRITE_SIZE"]["mean"] * 1024
    per[k] = {"launches_per_lde": n, "fetch_bytes_x2_per_launch": fb, "write_bytes_per_launch": wb, "mean_ms_under_profiler": f[k]["FETCH_SIZE"].get("mean_ms_under_profiler")}
    total += n * (fb + wb)
print(json.dumps({"workload": label, "per_kernel": per, "hbm_bytes_per_lde": total, "bytes_per_column_in_units_of_N": total / cols / N,
                  "algorithmic_bytes_per_column_in_units_of_N": 24.0}, indent=1))
