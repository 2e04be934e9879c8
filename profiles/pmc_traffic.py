"""Summarises two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes) into
per-kernel memory-side traffic per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --train-only --steps 1 --warmup 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --train-only --steps 1 --warmup 0
  python profiles/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of 1 KB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; Infinity-Cache hits are counted
(the counters sit on the L2's fabric side), so this is L2-miss traffic, an upper bound on HBM traffic."""
import collections
import csv
import glob
import json
import sys


def load(d, name):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main():
    fe, wr, out = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    rows = []
    for k in fe:
        n = len(fe[k])
        f = 2.0 * sum(fe[k]) / n * 1e3
        w = sum(wr.get(k, [0.0])) / max(1, len(wr.get(k, [0.0]))) * 1e3
        rows.append((k[0], k[1], n, f, w))
    rows.sort(key=lambda r: -(r[3] + r[4]) * r[2])
    with open(out + ".csv", "w") as fh:
        fh.write("kernel,grid_threads,launches,fetch_bytes_per_launch_x2_corrected,write_bytes_per_launch\n")
        for r in rows:
            fh.write('"%s",%d,%d,%.0f,%.0f\n' % r)
    def family(name, pred, what):
        sel = [r for r in rows if pred(r[0])]
        launches = sum(r[2] for r in sel)
        if not launches:
            return None
        return {"kernels": what, "launches": launches,
                "traffic_bytes_per_launch": sum((r[3] + r[4]) * r[2] for r in sel) / launches,
                "fetch_bytes_per_launch": sum(r[3] * r[2] for r in sel) / launches,
                "write_bytes_per_launch": sum(r[4] * r[2] for r in sel) / launches}
    summary = {
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH doubled (gfx950); counts L2-miss "
                "traffic including Infinity-Cache hits; per launch, averaged over the launches of the pass",
        "gemm_family": family("gemm", lambda k: "gemm_mfma" in k or "gemm_x256" in k,
                              "gemm_mfma_kernel + gemm_mfma_f32_kernel + gemm_x256_kernel"),
        "lstm_wide": family("wide", lambda k: "lstm_wide_fwd_kernel" in k or "lstm_wide_bwd" in k,
                            "lstm_wide_fwd_kernel + lstm_wide_bwd[_ps]_kernel (one launch = all decoder steps of one LSTM)"),
        "lstm_wide_fwd": family("widef", lambda k: "lstm_wide_fwd_kernel" in k, "lstm_wide_fwd_kernel"),
        "lstm_wide_bwd": family("wideb", lambda k: "lstm_wide_bwd" in k, "lstm_wide_bwd_kernel / lstm_wide_bwd_ps_kernel"),
        "attn_cluster": family("attn", lambda k: "attn_cluster_fwd_kernel" in k or "attn_cluster_bwd_kernel" in k,
                               "attn_cluster_fwd_kernel + attn_cluster_bwd_kernel"),
        "lstm_cluster": family("clu", lambda k: "lstm_cluster" in k, "lstm_cluster2_fwd_kernel + lstm_cluster2_bwd_kernel"),
    }
    json.dump(summary, open(out + ".json", "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
