#!/bin/bash
# GPU-box helper: HBM traffic (two PMC passes) of the replay kernels at the sizes of tools/micro_hbm.py, next to its timings.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hbm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/tools/micro_hbm.py 2>/dev/null | grep '^{' > $O/micro_hbm.json || exit 1
for L in ring arrays; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$L -- python3 $R/tools/micro_hbm.py $L > /dev/null 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$L -- python3 $R/tools/micro_hbm.py $L > /dev/null 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$L -- python3 $R/tools/micro_hbm.py $L > /dev/null 2>&1 || exit 1
  echo "== $L"
  python3 $R/tools/pmc_traffic.py $O/pmc_f_$L $O/pmc_w_$L $O/pmc_traffic_hbm_$L.json
done
