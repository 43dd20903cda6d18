# Phase traces of LONE workgroups (2 560 rows: one workgroup per CU) for the forward, critic backward and actor backward in the
# f16x2 mode -- what a launch of <= 256 tiles (c1, pre-training) is made of.  GPU box: bash tools/trace_small.sh (trace build)
set -e
cd $GRAFT_REPO_ROOT
MOBODY_TRACE=1 python - <<PY
import sys
sys.path.insert(0, "mobody-model-based-off-dynamics-offline-reinforcement-learning_amd/csrc")
import build
build.build(force=True, verbose=False)
PY
for mode in fwd critic actor; do echo "=== f16x2 $mode 2560 rows (one workgroup per CU)"; MOBODY_MFMA=f16x2 python tools/trace_mlp.py $mode 2560; done
