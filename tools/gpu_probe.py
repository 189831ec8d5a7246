#!/usr/bin/env python3
"""Timing probe used during development: device-resident synthetic slabs, the
*_dev entry points, hipEvent timings from ta_last_timing."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from transport_analysis_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="1000x10000,5000x20000,10000x20000")
    ap.add_argument("--mode", default="fft")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nwg", type=int, default=0)
    ap.add_argument("--bp", type=int, default=0)
    a = ap.parse_args()
    ctx = _lib.Context(0)
    if a.nwg:
        ctx.set_option("fft_nwg" if a.mode == "fft" else "direct_nwg", a.nwg)
    dev = torch.device("cuda:0")
    for case in a.cases.split(","):
        T, A = (int(x) for x in case.split("x"))
        D = 3
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        v = torch.randn((T, A, D), dtype=torch.float64, device=dev, generator=g)
        out = torch.zeros(T, dtype=torch.float64, device=dev)
        bp = torch.zeros((T, A), dtype=torch.float64, device=dev) if a.bp else None
        torch.cuda.synchronize()
        times = []
        for r in range(a.reps):
            if a.mode == "fft":
                ctx.vacf_fft_dev(v.data_ptr(), T, A, D, A * D, out.data_ptr(),
                                 bp.data_ptr() if a.bp else 0, A)
            elif a.mode == "direct":
                ctx.vacf_direct_dev(v.data_ptr(), T, A, D, A * D, out.data_ptr(),
                                    bp.data_ptr() if a.bp else 0, A)
            else:
                m = torch.ones(A, dtype=torch.float64, device=dev)
                ctx.helfand_msd_dev(v.data_ptr(), v.data_ptr(), m.data_ptr(), T, A, D, A * D, 1.0,
                                    out.data_ptr(), bp.data_ptr() if a.bp else 0, A)
            times.append(ctx.last_timing())
        # spot check a few lags against torch
        ts = (out / A).cpu()
        errs = []
        if a.mode != "helfand":
            for k in (0, 1, T // 3, T - 1):
                ref = (v[: T - k] * v[k:]).sum().item() / (T - k) / A
                errs.append(abs(ts[k].item() - ref))
        scale = abs(ts[0].item())
        tot = sorted(t for t, _ in times)[len(times) // 2]
        main_ms = sorted(m for _, m in times)[len(times) // 2]
        gb = T * A * D * 8 / 1e9
        print(f"{a.mode} T={T} A={A}: total {tot:.3f} ms, main {main_ms:.3f} ms, "
              f"{T*A/tot/1e6:.2f} Gpts/s, {gb/main_ms*1e3:.1f} GB/s algorithmic, "
              f"max spot err/scale {max(errs)/scale if errs else float('nan'):.2e}", flush=True)
        del v, out, bp


if __name__ == "__main__":
    main()
