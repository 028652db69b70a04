"""A/B of the fp32 sr_head.2 kernels inside the cfg-3 step (GPU box): matrix-core kernels vs the VALU kernels."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for flag in ("1", "0", "1", "0"):
    code = ("import sys; sys.path.insert(0, %r); sys.argv=['bench.py','--workload','seg','--steps','30','--no-cpu-baseline'];"
            "from rehrseg_amd import hip_backend as hb; hb.USE_THIN5_F32 = bool(%s); import runpy; runpy.run_path(%r, run_name='__main__')"
            % (root, flag, os.path.join(root, "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    import json
    j = json.loads(out)
    print("USE_THIN5_F32 =", flag, "ms_per_step", round(j["ms_per_step"], 2), "patches/s", round(j["value"], 2), flush=True)
