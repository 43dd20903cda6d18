# usage (GPU box): bash tools/probe/ab_pipe.sh "flags of variant 1" "flags of variant 2" ...   -- twin-Q forward A/B under graph replay
set -e
for F in "$@"; do
  bash tools/ab_build.sh mlp_fwd_pipe.hip "$F"
  TAG="$F" python tools/probe/micro_pipe.py
done
