#!/usr/bin/env python3
"""Kernel timeline of one synthetic workload on cuda:0 (the "timeline" option of the library).

    tools/timeline_case.py MODE T A [--bp] [--f32] [--reps N]       MODE: fft | direct | helfand
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["fft", "direct", "helfand"])
    ap.add_argument("T", type=int)
    ap.add_argument("A", type=int)
    ap.add_argument("--bp", action="store_true")
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch

    import bench
    from transport_analysis_amd import _lib

    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    c = bench.Case(torch, ctx, dev, args.mode, args.T, args.A, 3, 0, args.A * 3, bench.SEED + 4, args.bp, args.f32, False, False)
    ctx.set_option("timeline", 1)
    for r in range(args.reps + 1):
        c.step()
        torch.cuda.synchronize()
        tl = ctx.kernel_timeline()
        if r:
            print(f"{args.mode} {args.T} x {args.A} x 3 bp={args.bp} f32={args.f32}: total {sum(m for _, m in tl):.3f} ms  " +
                  "  ".join(f"{n} {m:.3f}" for n, m in tl))


if __name__ == "__main__":
    main()
