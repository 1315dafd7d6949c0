"""Skinny GEMM time with weights HBM-cold (12 rotating copies) vs Infinity-Cache-resident (the same copy again)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.bench_skinny as B
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    B.run(32, n, k, copies=12)
    B.run(32, n, k, copies=2, launches=12)   # 2 copies alternate: both stay in the 256 MiB cache, neither fits L2 twice
