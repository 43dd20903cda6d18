#!/bin/bash
# All bench lines of a round on the GPU box: bash tools/collect_bench.sh TAG
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; python tools/show_bench.py gpurun_out/${TAG}_bench_c2.json
for c in c1 c3 c4 c5; do python bench.py --config $c --no_cpu_baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err; python tools/show_bench.py gpurun_out/${TAG}_bench_$c.json; done
python bench.py --penalty par --no_cpu_baseline --no_mode_sweep > gpurun_out/${TAG}_bench_c2_par.json 2>/dev/null; python tools/show_bench.py gpurun_out/${TAG}_bench_c2_par.json
python bench.py --config pretrain > gpurun_out/${TAG}_bench_pretrain.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_pretrain.json'));print('pretrain', d['ms_per_step'], d['value'], d.get('gpu_over_cpu'))"
python tools/micro_hbm.py both > gpurun_out/${TAG}_micro_hbm.json 2>/dev/null
