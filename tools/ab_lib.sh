# usage (GPU box): bash tools/ab_lib.sh "flags A" "flags B" ...  -- rebuild the WHOLE library with each flag set and print c2 / c1 /
# pretrain step times (two c2 runs each) on one box
for F in "$@"; do
  MOBODY_EXTRA_FLAGS="$F" python mobody-model-based-off-dynamics-offline-reinforcement-learning_amd/csrc/build.py --force > /dev/null 2>&1 || exit 1
  for rep in 1 2; do
    python bench.py --no_cpu_baseline --no_mode_sweep > gpurun_out/ab_tmp.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); k=d['kernels']
print('[$F] c2 %.4f ms  fwd %.1f bwd %.1f wgrad %.1f us  rollout %.1f M/s' % (d['ms_per_step'], k['k_mlp3_fwd']['ms_per_step']*1e3, k['k_mlp3_bwd']['ms_per_step']*1e3, k['k_wgrad']['ms_per_step']*1e3, d['rollout_transitions_per_sec']/1e6))"
  done
  python bench.py --config c1 --no_cpu_baseline --no_mode_sweep > gpurun_out/ab_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('[$F] c1 %.4f ms' % d['ms_per_step'])"
  python bench.py --config pretrain --no_cpu_baseline > gpurun_out/ab_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('[$F] pretrain %.4f ms' % d['ms_per_step'])"
done
