"""Summarises one rocprofv3 PMC pass with the matrix-core counters into per-kernel MFMA utilisation.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \\
      -d gpurun_out/pmc_mfma -- python3 bench.py --no-cpu-baseline --train-only --steps 1 --warmup 0
  python3 profiles/pmc_mfma.py gpurun_out/pmc_mfma profiles/r04_pmc_mfma.csv

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the fraction of the chip's SIMD-cycles,
while the kernel ran, in which a SIMD's matrix pipe was busy (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles
per SIMD, 16 per v_mfma_f32_16x16x32_bf16; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs).  ROCm 7.2 ships no
gfx950 derived `MfmaUtil`; this is the gfx94x formula written out.  1.0 = every SIMD issuing MFMAs back to back = the
2.5 PFLOP/s dense bf16 peak, so for a bf16 kernel mfma_util x 2500 ~ its issued TFLOP/s."""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
    rows = []
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = mf / (gui / 8.0 * 256 * 4) if gui else 0.0
        rows.append((k, len(launches[k]), mf, c.get("SQ_BUSY_CYCLES", 0.0), gui, util))
    rows.sort(key=lambda r: -r[4])
    with open(out, "w") as fh:
        fh.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_util\n")
        for r in rows:
            fh.write('"%s",%d,%.0f,%.0f,%.0f,%.4f\n' % r)
    tot_gui = sum(r[4] for r in rows)
    tot_mf = sum(r[2] for r in rows)
    print("whole pass: mfma_util %.4f over %d kernels" % (tot_mf / (tot_gui / 8.0 * 256 * 4) if tot_gui else 0.0, len(rows)))
    for r in rows[:16]:
        print("%-100s launches %4d  gui %.3e  mfma_util %.4f" % (r[0][:100], r[1], r[4], r[5]))


if __name__ == "__main__":
    main()
