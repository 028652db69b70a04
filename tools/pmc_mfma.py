"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass (MI355X):

    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv ...

usage: python tools/pmc_mfma.py <counter_collection.csv> <kernel_trace.csv of the same pass> <out.json>

Units (MI355X_MICROARCH.md, 'Per-instruction cycle constants'): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over
all SIMDs (v_mfma_f32_32x32x2_f32 = 64 cycles, v_mfma_f32_32x32x16_bf16 = 32 cycles of one SIMD's matrix pipe);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  mfma_util = MFMA busy cycles / (1024 SIMDs x kernel cycles), kernel cycles =
GRBM_GUI_ACTIVE / 8; effective clock = kernel cycles / duration."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:80]


def main():
    cpath, tpath, out = sys.argv[1:4]
    dur = defaultdict(list)
    for r in csv.DictReader(open(tpath)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    cnt = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for r in csv.DictReader(open(cpath)):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    res = {}
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        c = cnt.get(k)
        if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        mean = lambda name: (sum(c[name].values()) / max(1, len(c[name]))) if name in c else None  # noqa: E731
        mfma, gui, t = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE"), sum(dur[k]) / len(dur[k])
        if not mfma or not gui:
            continue
        cyc = gui / 8.0
        res[k] = {"launches": len(dur[k]), "mean_us_in_this_pass": t * 1e6, "mfma_busy_cycles": mfma, "kernel_cycles": cyc,
                  "effective_clock_GHz": cyc / t / 1e9, "mfma_util": mfma / (1024.0 * cyc),
                  "sq_busy_cycles": mean("SQ_BUSY_CYCLES"), "sq_wave_cycles_x4": mean("SQ_WAVE_CYCLES")}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE",
               "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); profiled passes run at a "
                       "lower clock than un-profiled ones (MI355X_MICROARCH.md, DVFS give-back (2))", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in list(res.items())[:25]:
        print(f"{k:60s} {v['launches']:5d}x {v['mean_us_in_this_pass']:9.1f} us  mfma_util {v['mfma_util']:6.1%}  "
              f"clock {v['effective_clock_GHz']:.2f} GHz")


if __name__ == "__main__":
    main()
