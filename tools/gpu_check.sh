#!/bin/bash
# GPU-box helper: run the GPU parity suite, then a short bench and print a compact per-kernel summary.
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 || exit 1
for bs in ${BENCH_BS:-4096}; do
timeout -k 10 300 python bench.py --steps ${BENCH_STEPS:-100} --warmup 10 --no_cpu_baseline --batch_size $bs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['rows_per_step_per_gpu'], 'ms/step', round(d['ms_per_step'],3), 'grad-steps/s', round(d['grad_steps_per_sec'],1), 'rollout Mtr/s', round(d['rollout_transitions_per_sec']/1e6,2), {k:(round(v['ms_per_step'],3), round(v['tflops'],1)) for k,v in d['kernels'].items()})
"
done
