#!/usr/bin/env python3
"""A/B of the 2:4-sparse consumer of a pooled boundary (csrc/conv_sparse.h) against the dense split-bf16 kernel at the walk's
own shapes (320 tokens), operator level.  Run under `rocprofv3 --kernel-trace --stats` and read the per-kernel averages:
  conv_sparse_kernel                      the sparse launch (reps = 5)
  conv_igemm_kernel<2,4,4,2,2,1,true,..>  the dense 8-wave halo kernel on the EXPANDED tensor (what the walk ran before the
                                          compact interfaces: 2.77 / 2.88 ms; with the compact in-loop loader 2.43 / 2.51 ms,
                                          profiles/r03_layer_bench_bf16x3.txt)
Usage: python profiles/sparse_ab.py [block4_conv3|block3_conv3|both]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrp_imagecaptioning_amd.engine import op_conv, op_conv_pool_sparse  # noqa: E402

SHAPES = {"block4_conv3": (320, 14, 14, 512, 512), "block3_conv3": (320, 28, 28, 256, 256)}
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, (NB, Hp, Wp, Cin, Cout) in SHAPES.items():
    if which not in ("both", name):
        continue
    rs = np.random.RandomState(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    sc = torch.randn((NB, Hp, Wp, Cout), device="cuda", generator=g)
    pos = torch.randint(0, 4, (NB, Hp, Wp, Cout), device="cuda", generator=g, dtype=torch.uint8)
    w = np.abs(rs.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cout)).astype(np.float32)
    gate = torch.rand((NB, 2 * Hp, 2 * Wp, Cin), device="cuda", generator=g)
    S = torch.zeros((NB, 2 * Hp, 2 * Wp, Cout), device="cuda")
    for p in range(4):
        S[:, (p >> 1)::2, (p & 1)::2, :] = torch.where(pos == p, sc, torch.zeros_like(sc))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = op_conv_pool_sparse(sc, pos, w, gate, reps=5)
    t1 = time.perf_counter()
    for diag in [int(d) for d in os.environ.get("SP_DIAG", "").split(",") if d]:     # measurement variants (results meaningless), 3 launches each
        op_conv_pool_sparse(sc, pos, w, gate, reps=3 | (diag << 8))
    dense = None
    for _ in range(3):
        dense = op_conv(S, w, None, gate, 2, 9, split_bf16=True)
    err = float((got.double() - dense.double()).abs().sum() / dense.double().abs().sum())
    flop = 2.0 * NB * 4 * Hp * Wp * 9 * Cin * Cout
    print("%s: n=%d %dx%d windows, %d -> %d channels: sparse vs dense kernel relative L1 %.3g; %.1f algorithmic GFLOP per launch"
          % (name, NB, Hp, Wp, Cout, Cin, err, flop / 1e9), flush=True)
