"""The Tacotron-1 block of bench.py alone (BASELINE config 1's model at the benchmark shape)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

a = argparse.Namespace(batch=32, t_in=160, t_out=1000, dtype=sys.argv[1] if len(sys.argv) > 1 else "mixed")
print(json.dumps(bench.taco1_bench(a, None), indent=1))
