set -e
cd $GRAFT_REPO_ROOT
for M in 2048 4096 16384; do timeout -k 10 200 python tools/debug/pingpong_ab.py $M 4096:14336,6144:4096,4096:4096,28672:4096 2>&1 | grep "^M=\|^silu" | cut -c1-330; done
timeout -k 10 300 python tools/debug/pingpong_ab.py 16384 8192:8192,10240:8192,8192:28672 2>&1 | grep "^M=" | cut -c1-330
