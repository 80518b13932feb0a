"""Per-kernel HBM traffic and MFMA utilisation of one training step from rocprofv3 counter passes
(MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE doubled on gfx950; values are KB; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  tools/evidence_r0N.sh runs the
passes; then:

  python tools/pmc_report.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma <kernel_stats.csv> profiles/r0N_pmc_traffic.json

Durations come from the --kernel-trace --stats pass of the graph-replayed bench (average per launch)."""
import csv
import glob
import json
import os
import statistics
import sys

KERNELS = ["vocab_fwd_dh_kernel", "vocab_fwd_combine_kernel", "vocab_dw_kernel", "vocab_dw_reduce_kernel", "gemm_nt_kernel",
           "gemm_nt_ct_kernel", "gru_fwd_kernel", "gru_bwd_kernel", "block_fwd_kernel", "block_fwd_cluster_kernel",
           "block_bwd_dx_kernel", "block_bwd_dx_cluster_kernel",
           "linear_f32_group_kernel", "gather_kernel", "segment_sum_kernel", "adam_kernel", "sumsq_kernel",
           "dls_level_fwd_kernel", "dls_level_bwd_kernel", "dropout_kernel", "cast_pack_vec4_kernel"]
VOCAB = KERNELS[:4]


def key_of(name):
    for k in KERNELS:
        if k + "(" in name or k + "<" in name:
            return k
    return None


def collect(d, counter):
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    vals = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = key_of(r["Kernel_Name"])
        if k:
            vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return vals


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    mfma, gui = collect(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"), collect(sys.argv[3], "GRBM_GUI_ACTIVE")
    stats = {}
    for r in csv.DictReader(open(sys.argv[4])):
        k = key_of(r["Name"])
        if k:
            s = stats.setdefault(k, [0, 0.0])
            s[0] += int(r["Calls"])
            s[1] += float(r["TotalDurationNs"])
    steps = stats["adam_kernel"][0]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three passes) -- python3 "
                     "bench.py --cpu-baseline 0 --graph 0 --steps 3 --warmup 1 --profile-steps 1; medians per launch; FETCH_SIZE "
                     "doubled (gfx950), KB x 1024; durations: average per launch of the --kernel-trace --stats pass of the "
                     "graph-replayed bench",
           "kernels": {}}
    total = 0
    for k in KERNELS:
        if k not in stats:
            continue
        calls, ns = stats[k]
        e = {"launches_per_step": round(calls / steps, 2), "avg_us": round(ns / calls / 1e3, 2)}
        if k in fetch or k in write:
            f, w = statistics.median(fetch.get(k, [0.0])), statistics.median(write.get(k, [0.0]))
            b = int((2 * f + w) * 1024)
            e.update(FETCH_SIZE_KB_median=f, WRITE_SIZE_KB_median=w, hbm_bytes_per_launch=b,
                     achieved_GBps=round(b / (ns / calls), 1), frac_of_8TBps=round(b / (ns / calls) / 8000.0, 3))
            if k in VOCAB:
                total += b * calls / steps
        if k in mfma and k in gui:
            busy, act = statistics.median(mfma[k]), statistics.median(gui[k])
            if act > 0:
                e.update(SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=act, mfma_busy_frac=round(busy / (1024.0 * act / 8.0), 4))
        out["kernels"][k] = e
    out["vocab_group_hbm_bytes_per_step"] = int(total)
    json.dump(out, open(sys.argv[5], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
