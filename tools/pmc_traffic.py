#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots").

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv \
         --kernel gemm_bf16_kernelILi2ELi4ELi4ELi2ELb1ELb1ELi128EDF16b --grid 786432 --key "<bench roofline.kernel string>" --out profiles/pmc_traffic.json

Corrections (guide, "HBM" section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
(16 B / lane) coalesced reads, which is what every kernel here issues, so traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.
"""
import argparse
import csv
import json
import os


def mean_counter(path, counter, kernel, grid):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == str(grid)):
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for kernel~{kernel} grid={grid} in {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--kernel", required=True, help="substring of the (mangled) kernel name")
    ap.add_argument("--grid", type=int, default=None, help="total grid size (work-items) to pick one shape of a template")
    ap.add_argument("--key", required=True, help="bench.py roofline.kernel string this measurement belongs to")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    fetch_kib, nf = mean_counter(a.fetch_csv, "FETCH_SIZE", a.kernel, a.grid)
    write_kib, nw = mean_counter(a.write_csv, "WRITE_SIZE", a.kernel, a.grid)
    rec = {"fetch_size_kib_raw": round(fetch_kib, 1), "write_size_kib_raw": round(write_kib, 1), "dispatches": [nf, nw],
           "read_bytes": int(2 * fetch_kib * 1024), "write_bytes": int(write_kib * 1024),
           "traffic": int(2 * fetch_kib * 1024 + write_kib * 1024),
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), both counters KiB"}
    print(json.dumps({a.key: rec}, indent=1))
    if a.out:
        cur = {}
        if os.path.exists(a.out):
            cur = json.load(open(a.out))
        cur[a.key] = rec
        json.dump(cur, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
