"""HBM traffic and achieved bandwidth per kernel from three rocprofv3 passes of ONE command (MI355X):

    rocprofv3 --kernel-trace --output-format csv ...            (durations, un-perturbed by counters)
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv ...

usage: python tools/pmc_hbm.py <kernel_trace.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
       [--algo bench_stream.json]

Units and corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so
fetched bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.  Infinity-Cache
hits are counted in both, so "hbm_bytes" is traffic at the L2's memory side (an upper bound of DRAM traffic).
bandwidth = (2 x FETCH + WRITE) / mean duration of the same kernel in the counter-free pass; peak 8 TB/s (spec),
6.29 TB/s (measured float4 copy)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(?:I|E)", name)
    if m:  # mangled template instance: kernel name + element type
        return m.group(1) + ("<bf16>" if "DF16b" in name else "<f32>")
    return re.sub(r"\(.*$", "", name)


def durations(path):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return per


def counter(path, which):
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == which:
            per[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per


def main():
    trace, fpath, wpath, out = sys.argv[1:5]
    algo = None
    if "--algo" in sys.argv:
        algo = json.load(open(sys.argv[sys.argv.index("--algo") + 1]))
    dur, fe, wr = durations(trace), counter(fpath, "FETCH_SIZE"), counter(wpath, "WRITE_SIZE")
    res = {}
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if k not in fe and k not in wr:
            continue
        n = len(dur[k])
        f_kib = sum(fe.get(k, {}).values()) / max(1, len(fe.get(k, {})))
        w_kib = sum(wr.get(k, {}).values()) / max(1, len(wr.get(k, {})))
        t = sum(dur[k]) / n
        b = (2 * f_kib + w_kib) * 1024
        res[k] = {"launches": n, "mean_us": t * 1e6, "fetch_size_kib_per_launch": f_kib, "write_size_kib_per_launch": w_kib,
                  "hbm_bytes_per_launch": b, "achieved_TBps": b / t / 1e12, "frac_of_8TBps": b / t / 8e12,
                  "frac_of_6p29TBps": b / t / 6.29e12}
    doc = {"source": "rocprofv3 --kernel-trace (durations) / --pmc FETCH_SIZE / --pmc WRITE_SIZE: three separate passes of "
                     "the same command on MI355X",
           "units": "KiB; fetched bytes = 2 x FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; "
                    "Infinity-Cache hits are included", "kernels": res}
    if algo is not None:   # algorithmic bytes of the same launches (tools/bench_stream.py --json)
        agg = defaultdict(lambda: [0.0, 0.0])
        for r in algo["rows"]:
            agg[r["kernel"]][0] += r["algorithmic_bytes"]
            agg[r["kernel"]][1] += r["us"] * 1e-6
        doc["algorithmic"] = {k: {"bytes_all_shapes": v[0], "seconds_all_shapes": v[1], "TBps": v[0] / v[1] / 1e12,
                                  "frac_of_8TBps": v[0] / v[1] / 8e12} for k, v in agg.items()}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in list(res.items())[:40]:
        print(f"{k:44s} {v['launches']:5d}x {v['mean_us']:9.1f} us  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"{v['achieved_TBps']:5.2f} TB/s ({v['frac_of_8TBps']:5.1%} of 8)")


if __name__ == "__main__":
    main()
