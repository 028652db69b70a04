"""A/B of one hip_backend switch inside a bench.py workload (GPU box):
    python tools/ab_flag.py WINO_BAND_MAJOR flavr [extra bench args]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flag, workload, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
for val in ("True", "False", "True", "False"):
    argv = ["bench.py", "--workload", workload, "--steps", "30", "--no-cpu-baseline", "--no-kernel-timing"] + extra
    code = ("import sys; sys.path.insert(0, %r); sys.argv=%r;"
            "from rehrseg_amd import hip_backend as hb; hb.%s = %s; import runpy; runpy.run_path(%r, run_name='__main__')"
            % (root, argv, flag, val, os.path.join(root, "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    j = json.loads(out)
    print(flag, "=", val, workload, "ms_per_step", round(j["ms_per_step"], 2), "patches/s", round(j["value"], 2), flush=True)
