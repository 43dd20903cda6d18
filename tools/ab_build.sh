# usage: bash tools/ab_build.sh FILE.hip "-DX=1 -DY=2"   (rebuild one translation unit with extra flags and relink; GPU-box A/B aid)
B=mobody-model-based-off-dynamics-offline-reinforcement-learning_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function $2 -c $B/$1 -o $B/build/${1%.hip}.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmobody_hip.so $B/build/*.o
