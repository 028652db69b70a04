"""A/B of several builds of librehrseg_hip.so inside one bench.py workload (GPU box):
    python tools/ab_libs.py workload lib1.so lib2.so ...   ("-" = the in-tree library); two rounds, alternating."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, libs = sys.argv[1], sys.argv[2:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["REHRSEG_HIP_LIB"] = os.path.abspath(lib)
        else:
            env.pop("REHRSEG_HIP_LIB", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", w, "--steps", "30", "--no-cpu-baseline",
                              "--no-kernel-timing"], capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        print(w, lib, "ms_per_step", round(j["ms_per_step"], 2), "median", round(j["step_ms"]["median"], 2), flush=True)
