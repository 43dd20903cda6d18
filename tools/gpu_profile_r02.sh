#!/bin/bash
# GPU-box helper for the round-2 profiles: bench lines of every config, the kernel-trace summary of the default bench and
# the two PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; no trace domains next to --pmc on this pool).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c2 c1 c3 c4 c5; do  # default --mfma bf16x3; the other modes ride along in other_mfma_modes
  timeout -k 10 280 python3 $R/bench.py --config $c $( [ $c = c2 ] || echo --no_cpu_baseline ) 2>/dev/null | grep '^{' > $O/bench_$c.json || exit 1
  echo "bench $c done"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_mode_sweep --graph 0 2>/dev/null | grep '^{' > $O/bench_under_rocprof.json || exit 1
echo "kernel trace done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2>&1 || exit 1
echo "pmc fetch done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2>&1 || exit 1
echo "pmc write done"
python3 $R/tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json | head -30
