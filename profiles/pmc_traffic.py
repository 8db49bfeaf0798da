#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter-collection CSVs (separate passes) -> per-kernel HBM bytes per launch.

    python profiles/pmc_traffic.py FETCH.csv WRITE.csv B L F out.json

Counter unit is KiB.  On gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so the read side is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is taken as is."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "ptr::" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def source_hash():
    """The hash bench.py checks before trusting this file: sha256 over the HIP sources the .so was built from."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_hash()


def main(fetch_csv, write_csv, B, L, F, out):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB_raw": round(fk, 1), "WRITE_SIZE_KiB_raw": round(wk, 1),
                      "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024))}
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (bench.py --steps 6 --warmup 2 --no-cpu-baseline, 1xMI355X). "
                   "Counter unit is KiB. On gfx950 FETCH_SIZE counts 128-B requests as 64 B, so the read side is doubled "
                   "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.",
           "config": {"queries_per_gpu_per_step": int(B), "list_len": int(L), "features": int(F)},
           "kernel_source_hash": source_hash(), "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main(*sys.argv[1:7])
