set -e
cd $GRAFT_REPO_ROOT
B=mobody-model-based-off-dynamics-offline-reinforcement-learning_amd/csrc
for W in 2 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -DFWD_F16_WAVES=$W -c $B/mlp_fwd_bf.hip -o $B/build/mlp_fwd_bf.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmobody_hip.so $B/build/*.o
  echo "== FWD_F16_WAVES=$W"
  python tools/micro_fwd_bf.py 10240 15360 2>&1 | grep rows | sed 's/| bf16 .*| f16x2/| f16x2/'
  python bench.py --mfma f16x2 --no_cpu_baseline --no_mode_sweep > gpurun_out/ab_occ_$W.json 2>/dev/null; python tools/show_bench.py gpurun_out/ab_occ_$W.json
done
