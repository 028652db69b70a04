"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Units follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE counts
64 B per 128-B request for wide coalesced reads, so fetched bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact.
"""
import csv, json, re, sys
from collections import defaultdict

FAMILIES = [("wino_conv_big", r"wino_conv_big8_kernel"), ("wino_conv_w32", r"wino_conv_w32p_kernel"),
            ("wino_conv_small", r"wino_conv_kernel"), ("wino22_conv", r"wino22_conv_kernel"),
            ("wino22_wgrad", r"wino22_wgrad_kernel"), ("wino_wgrad", r"wino_wgrad_kernel"), ("halo_conv", r"halo_conv_kernel"),
            ("gather_gemm", r"gather_gemm(_multi)?_kernel"), ("wgrad_slab", r"wgrad_kernel<"),
            ("wgrad_brick", r"wgrad_brick_kernel")]


def read(path, counter):
    per = defaultdict(lambda: defaultdict(float))   # family -> dispatch id -> value
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            for fam, pat in FAMILIES:
                if re.search(pat, name):
                    per[fam][row["Dispatch_Id"]] += float(row["Counter_Value"])
                    break
    return per


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 "
                 "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing, MI355X",
       "units": "FETCH_SIZE and WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md FETCH_SIZE counts 64 B per "
                "128-B request on gfx950 for wide coalesced reads, so fetched bytes = 2 * FETCH_SIZE; WRITE_SIZE is exact",
       "kernels": {}}
for fam, _ in FAMILIES:
    if fam not in fetch:
        continue
    n = len(fetch[fam])
    fk = sum(fetch[fam].values()) / n
    wk = sum(write.get(fam, {}).values()) / max(1, len(write.get(fam, {})))
    out["kernels"][fam] = {"launches_sampled": n, "fetch_size_kib_per_launch": fk, "write_size_kib_per_launch": wk,
                           "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
