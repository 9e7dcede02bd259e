export TMPDIR=/tmp
DVT_TAG=r06s2 bash tools/gpu.sh "test:stage2" 2>&1 | tail -4
for b in 1 0 1 0; do DVT_S2_BIG=$b python tools/bench_stage2.py 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/DVT_S2_BIG=$b: /"; done
