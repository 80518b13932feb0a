"""HBM traffic of the vocabulary GEMM group from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section:
separate --pmc passes; FETCH_SIZE is doubled on gfx950).  Run on the GPU box from /tmp:

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --cpu-baseline 0 --graph 0 --steps 3 --warmup 1 --profile-steps 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py (same arguments)

then here:  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json"""
import csv
import glob
import json
import os
import statistics
import sys

def collect(d, counter):
    """Kernels of the vocabulary group are recognised by launch order (the persistent tile walk gives every large GEMM
    the same grid): EpiSoftmax = forward; after each EpiProb the next two EpiStore launches are dH and dW and the next two
    splitk_reduce launches their slab reductions."""
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = {}
    stores, reduces = [], []
    for r in rows:
        name = r["Kernel_Name"]
        k = None
        if "EpiSoftmax" in name:
            k = "vocab_fwd"
        elif "EpiProb" in name:
            k = "vocab_bwd_prob"
            stores, reduces = ["vocab_dH", "vocab_dW"], ["vocab_dH_reduce", "vocab_dW_reduce"]
        elif "EpiStore" in name and stores:
            k = stores.pop(0)
        elif "splitk_reduce" in name and reduces:
            k = reduces.pop(0)
        if k:
            vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in vals.items()}, f


def main():
    fetch, f1 = collect(sys.argv[1], "FETCH_SIZE")
    write, f2 = collect(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --cpu-baseline 0 "
                     "--graph 0 --steps 3 --warmup 1 --profile-steps 1; FETCH_SIZE doubled (gfx950 correction); KB -> bytes x1024",
           "kernels": {}}
    total = 0
    for k in sorted(set(fetch) | set(write)):
        b = int((2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)
        out["kernels"][k] = {"FETCH_SIZE_KB_median": fetch.get(k), "WRITE_SIZE_KB_median": write.get(k), "hbm_bytes_per_launch": b}
        total += b
    out["vocab_group_hbm_bytes_per_step"] = total
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
