# A/B of the second HIP streams of a training step (weight gradients; frozen teacher) on one box:
# bash tools/ab_streams.sh > gpurun_out/ab_streams.txt
mkdir -p gpurun_out
for w in flavr seg flavr_ref cfg4 cfg5; do
for k in "" "--no-wgrad-stream" "--no-teacher-stream" "--no-wgrad-stream --no-teacher-stream"; do
case "$w$k" in flavr*teacher*|seg*teacher*) continue;; esac
python3 bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing $k > gpurun_out/ab_x.json 2>/dev/null || exit 1
python3 - "$w" "$k" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/ab_x.json").read().strip().splitlines()[-1])
print("%-10s %-44s mean %.2f  median %.2f ms" % (sys.argv[1], sys.argv[2] or "(default: both streams)", r["ms_per_step"], r["step_ms"]["median"]))
PY
done
done
