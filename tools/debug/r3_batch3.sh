#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/r3_full_t2.log 2>&1; tail -5 $O/r3_full_t2.log
