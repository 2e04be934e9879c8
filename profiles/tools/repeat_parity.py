#!/usr/bin/env python3
"""Runs one strict gradient-parity case of tests/test_taco2_gpu.py repeatedly in one process and reports every
tensor that leaves its bound (numeric flake hunting; no kernel is launched that the test suite does not launch).
Usage: python profiles/tools/repeat_parity.py [count] [N Ti To]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_taco2_gpu as T  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    shape = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (33, 9, 10)
    bad = 0
    for i in range(count):
        try:
            T.test_taco2_fp32_forward_backward_matches_oracle(torch.device("cuda:0"), shape)
        except AssertionError as e:
            bad += 1
            print("run %d FAILED: %s" % (i, str(e)[:600]), flush=True)
        torch.cuda.empty_cache() if i % 3 == 0 else None
    print("%d of %d runs failed" % (bad, count))


if __name__ == "__main__":
    main()
