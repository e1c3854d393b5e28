"""median step time of the bench workload (no CPU baseline, no roofline replay) -- for A/B runs through scripts/lib_multi.py"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-roofline", "--steps", "16", "--warmup", "3", "--no-dead-row-line"] + sys.argv[1:],
                   capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
g = d.get("gpu") or {}
print(f"ms_per_step median {d['ms_per_step']:.2f} mean {d['ms_per_step_mean']:.2f} min/max {d['ms_per_step_min_max']} | sclk {g.get('sclk_mhz_median')} MHz, "
      f"{g.get('power_w_mean')} W mean | {d.get('energy_j_per_step')} J/step", flush=True)
