#!/usr/bin/env python3
"""Per-layer timing of the fine-tune step's conv weight-gradient product (csrc/train_gemm.h, lrp_op_conv_wgrad) at
VGG16 shapes, batch 32: algorithmic TFLOP/s against the fp32 matrix peak (157.3 TFLOP/s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from lrp_imagecaptioning_amd.engine import op_conv_wgrad
    from lrp_imagecaptioning_amd.synthetic import VGG16_CFG
    B = int(os.environ.get("B", 32))
    bf16 = os.environ.get("BF16", "0") != "0"          # BF16=1: lrp_op_conv_wgrad_bf16 (csrc/train_gemm_bf16.h), peak 2500 TFLOP/s
    peak = 2500.0 if bf16 else 157.3
    hw, tot_ms, tot_fl = 224, 0.0, 0.0
    for name, cin, cout, pool in VGG16_CFG:
        x = torch.randn((B, hw, hw, cin), device="cuda")
        dz = torch.randn((B, hw, hw, cout), device="cuda")
        op_conv_wgrad(x, dz, bf16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3
        e0.record()
        for _ in range(n):
            op_conv_wgrad(x, dz, bf16)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * hw * hw * 9 * cin * cout
        print("  wgrad %-14s %7.3f ms  %8.1f GFLOP  %6.1f TF/s (%4.1f%% of %.1f)" % (name, ms, fl / 1e9, fl / ms / 1e9, fl / ms / 1e9 / peak * 100, peak))
        tot_ms += ms; tot_fl += fl
        del x, dz
        if pool:
            hw //= 2
    print("  wgrad total      %7.3f ms  %8.1f GFLOP  %6.1f TF/s (incl. the bias column sums and the scratch allocation of the op entry)"
          % (tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))


if __name__ == "__main__":
    main()
