#!/usr/bin/env python3
"""Phase timings inside the persistent wide-cell LSTM kernels (csrc/lstm_wide.hip) at the benchmark shape:
NS_WIDE_TRACE=1 makes workgroup 0 stamp the 100 MHz clock at the phase boundaries of every step; this prints the mean
duration of each phase.  Usage: python profiles/wide_trace.py > profiles/r02_wide_trace.txt"""
import os
import sys

os.environ["NS_WIDE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402



def report(name, work, S):
    tr = work[64:64 + 256 * 8 * 2].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64)
    r, nx = tr[4:min(S, 256) - 1], tr[5:min(S, 256)]
    us = lambda x: x.mean() * 1e-2
    print("%s: %.2f us per step" % (name, us(nx[:, 0] - r[:, 0])))
    print("  sweeper wave 0: sweep until every piece is new      %6.2f us  (%.2f passes)" % (us(r[:, 1] - r[:, 0]), r[:, 5].mean()))
    print("  sweeper wave 0: MFMA, partials to LDS, barrier       %6.2f us" % us(r[:, 2] - r[:, 1]))
    print("  cell wave: barrier -> state published                %6.2f us" % us(r[:, 4] - r[:, 3]))
    print("  publish -> the NEXT step's sweep complete            %6.2f us" % us(nx[:, 1] - r[:, 4]))
    print("    publish -> wave 0 has seen all its producers       %6.2f us" % us(nx[:, 6] - r[:, 4]))
    print("    -> its last pieces loaded and checked              %6.2f us" % us(nx[:, 1] - nx[:, 6]))

def report_ps(name, work, S):
    """lstm_wide_bwd_ps_kernel (partial-sum exchange): [0] step start, [1] every polled granule carries this step's tag
    ([5] passes beyond the first), [2] behind barrier A, [3] behind barrier B (cell update done), [4] products done and
    published."""
    tr = work[64:64 + 256 * 8 * 2].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64)
    r, nx = tr[4:min(S, 256) - 1], tr[5:min(S, 256)]
    us = lambda x: x.mean() * 1e-2
    print("%s (partial-sum exchange): %.2f us per step" % (name, us(nx[:, 0] - r[:, 0])))
    print("  product wave 0: polls until its granules are new     %6.2f us  (%.2f extra passes)" % (us(r[:, 1] - r[:, 0]), r[:, 5].mean()))
    print("  sums to LDS + barrier A                              %6.2f us" % us(r[:, 2] - r[:, 1]))
    print("  cell update (cell waves) + barrier B                 %6.2f us" % us(r[:, 3] - r[:, 2]))
    print("  MFMA + publish                                       %6.2f us" % us(r[:, 4] - r[:, 3]))
    print("  publish -> the NEXT step's granules all in           %6.2f us" % us(nx[:, 1] - r[:, 4]))
    print("  (step start -> wave 7's polls done %.2f us, -> the cell waves reach barrier A %.2f us)" % (us(r[:, 7] - r[:, 0]), us(r[:, 6] - r[:, 0])))


def main():
    hp = hparams_mod.load("taco2")
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=1234)
    inputs, lengths, mel, lin = bench.synthetic_batch(hp, 32, 160, 1000, 1234)
    m.add_optimizer(0)
    for _ in range(3):
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
    torch.cuda.synchronize()
    m.check_status()
    for k in sorted(m._bufs):
        if k.startswith("lstm_wide_work_"):
            if k.endswith("_bwd") and os.environ.get("NS_WIDE_PS", "1") != "0":
                report_ps(k[len("lstm_wide_work_"):], m._bufs[k], 200)
            else:
                report(k[len("lstm_wide_work_"):], m._bufs[k], 200)


if __name__ == "__main__":
    main()
