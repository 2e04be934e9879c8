#!/usr/bin/env python3
"""Per-kernel LDS bank-conflict share from one rocprofv3 PMC pass:
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --no-cpu-baseline --train-only --steps 1 --warmup 0
  python profiles/tools/lds_conflicts.py gpurun_out/pmc_lds
SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        calls[r["Kernel_Name"]] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_LDS_BANK_CONFLICT"])
print("%-90s %8s %14s %14s %6s" % ("kernel", "launches", "conflict_cyc", "lds_active_cyc", "share"))
for k, v in rows[:40]:
    c, a = v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_IDX_ACTIVE"]
    print("%-90s %8d %14.0f %14.0f %6.2f" % (k[:90], calls[k], c, a, c / a if a else 0.0))
