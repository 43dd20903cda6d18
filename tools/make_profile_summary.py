"""profiles/<tag>_summary.md from a rocprofv3 kernel_stats CSV and the bench lines of the same build.
usage: make_profile_summary.py TAG TITLE kernel_stats.csv bench_under_rocprof.json bench_unprofiled.json"""
import csv, json, sys
tag, title, stats, prof_json, bench_json = sys.argv[1:6]
rows = list(csv.DictReader(open(stats)))
d, u = json.load(open(prof_json)), json.load(open(bench_json))
c = u.get("cpu_baseline") or {}
out = [f"# {title}", "",
       "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no_cpu_baseline --graph 0`",
       "(130 train() steps at bs=4096 -> N=10240 rows, plus 7 rollout steps of 50 000 rows and the step-1 refresh).",
       f"bench line of the same run (under the profiler, eager): {d['ms_per_step']:.3f} ms/step; un-profiled default run: "
       f"profiles/{tag}_bench_bs4096.json ({u['ms_per_step']:.3f} ms/step, {u['value']/1e6:.1f} M transitions/s, "
       f"{u['grad_steps_per_sec']:.0f} grad-steps/s, rollout {u['rollout_transitions_per_sec']/1e6:.1f} M transitions/s"
       + (f", cpu_baseline {c['grad_steps_per_sec']:.1f} grad-steps/s on {c['cores']} threads)." if c else ")."), "",
       "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:22]:
    out.append(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
out += ["", "In-process HIP-event timing of the same run (bench.py `kernels`): " + ", ".join(
    f"{n} {v['ms_per_step']/v['launches_per_step']*1e3:.1f} us/launch ({v['tflops']:.1f} TF)" for n, v in d["kernels"].items()) + ".", ""]
open(f"profiles/{tag}_summary.md", "w").write("\n".join(out))
print("\n".join(out))
