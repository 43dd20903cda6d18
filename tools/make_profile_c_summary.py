"""profiles/r02_c_summary.md from the files tools/gpu_profile_r02.sh produced (copied to profiles/r02_c_*)."""
import csv, json
P = 'profiles/r02_c_'
b = {c: json.loads(open(f'{P}bench_{c}.json').read().strip().splitlines()[-1]) for c in ['c1', 'c2', 'c3', 'c4', 'c5']}
u = b['c2']; d = json.loads(open(P + 'bench_under_rocprof.json').read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(P + 'kernel_stats_bench_c2.csv')))
cb = u.get('cpu_baseline') or {}
o = u['other_mfma_modes']
out = ["# Round 2, profile C (final round-2 build): unmasked scalar-addressed weight-gradient loop (split precision in every bf16 mode), "
       "row-interleaved replay ring, K-split wide transition heads, counter-free captured step", "",
       "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_mode_sweep --graph 0`",
       "(bench default `--mfma bf16x3`; 130 train() steps at bs=4096 -> N = 10 240 rows, 7 rollout steps of 50 000 rows and the step-1 refresh).",
       f"Un-profiled default run (`profiles/r02_c_bench_c2.json`): **{u['ms_per_step']:.3f} ms/step** incl. the refresh amortised at 1/5000 ({u['grad_steps_per_sec']:.0f} grad-steps/s), "
       f"{u['value']/1e6:.1f} M minibatch transitions/s, rollout {u['rollout_transitions_per_sec']/1e6:.1f} M transitions/s; same run, other modes (100 steps each, refresh excluded): "
       + ", ".join(f"{k} {v['ms_per_step_refresh_excluded']:.3f} ms" for k, v in o.items()) + ".",
       "Box-to-box spread of one build is about +-4 % (the same build read 0.296 ms/step, c3 0.865, c4 0.646 on another box two hours earlier)."]
if cb:
    out.append(f"CPU oracle ({cb.get('cores')} threads, {cb.get('cpu_model', '')}): {cb.get('grad_steps_per_sec', 0):.1f} grad-steps/s at these shapes.")
out += ["", "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:20]:
    out.append(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
out += ["", "In-process HIP-event timing of the same run (bench.py `kernels`, TFLOP/s are fp32-equivalent): " + ", ".join(
    f"{n} {v['ms_per_step']/v['launches_per_step']*1e3:.1f} us/launch ({v['tflops']:.1f} TF)" for n, v in d["kernels"].items()) + ".", ""]
out += ["All configs, un-profiled (`profiles/r02_c_bench_c*.json`), default bf16x3 with the other modes' step time (ms, refresh excluded) from the same run:", "",
        "| config | ms/step | grad-steps/s | minibatch Mtr/s | rollout Mtr/s | k_dyn_fwd TF | f32 | bf16x2 | bf16 |", "|---|---|---|---|---|---|---|---|---|"]
for c, x in b.items():
    oo = x['other_mfma_modes']
    out.append(f"| {c} | {x['ms_per_step']:.3f} | {x['grad_steps_per_sec']:.0f} | {x['value']/1e6:.1f} | {x['rollout_transitions_per_sec']/1e6:.1f} | {x['kernels']['k_dyn_fwd']['tflops']:.0f} | "
               + " | ".join(f"{oo[k]['ms_per_step_refresh_excluded']:.3f}" for k in ('f32', 'bf16x2', 'bf16')) + " |")
out += ["", open('profiles/r02_c_summary_tail.md').read().rstrip()]
open('profiles/r02_c_summary.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out[:12]))
