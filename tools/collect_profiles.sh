#!/bin/bash
# Round profile set on the GPU box: bash tools/collect_profiles.sh TAG   (writes gpurun_out/prof_TAG/*, summaries to gpurun_out/TAG_*)
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_mode_sweep --graph 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > gpurun_out/${TAG}_bench_under_rocprof.json 2> $O/kt.err && cp $O/kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_bench_c2.csv
echo "kernel trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc_m -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/pmc_m.err && python tools/pmc_mfma.py $O/pmc_m gpurun_out/${TAG}_pmc_mfma.json | tail -12
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w gpurun_out/${TAG}_pmc_traffic.json c2 f16x2 | head -14
echo "pmc traffic done"
