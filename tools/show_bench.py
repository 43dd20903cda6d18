"""Print the headline numbers of a bench.py JSON line (diagnostic)."""
import json, sys
for p in sys.argv[1:]:
    d = json.load(open(p))
    k = d["kernels"]
    print(f"{p}: {d['dtype']} ms/step {d['ms_per_step']:.4f}  refresh-excluded {1e3 / d['grad_steps_per_sec_refresh_excluded']:.4f}  "
          f"rollout {d['rollout_transitions_per_sec'] / 1e6:.1f} M/s  roofline {d['roofline']['kernel']} frac {d['roofline']['frac']:.3f}")
    print("   " + "  ".join(f"{n}: {v['ms_per_step'] * 1e3:.1f} us/step ({v['launches_per_step']:.0f} launches, {v['tflops']:.0f} TF)" for n, v in k.items()))
    if d.get("other_mfma_modes"):
        print("   other modes: " + "  ".join(f"{m}: {v['ms_per_step_refresh_excluded']:.4f}" for m, v in d["other_mfma_modes"].items()))
