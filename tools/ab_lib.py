"""A/B of two builds of librehrseg_hip.so inside bench.py workloads (GPU box):
    python tools/ab_lib.py <other.so> workload [workload ...]
Alternates the in-tree library and <other.so> (REHRSEG_HIP_LIB), two runs each."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other, workloads = os.path.abspath(sys.argv[1]), sys.argv[2:]
for w in workloads:
    for lib in ("", other, "", other):
        env = dict(os.environ)
        if lib:
            env["REHRSEG_HIP_LIB"] = lib
        else:
            env.pop("REHRSEG_HIP_LIB", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", w, "--steps", "30", "--no-cpu-baseline"],
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        print(w, "other" if lib else "in-tree", "ms_per_step", round(j["ms_per_step"], 2), flush=True)
