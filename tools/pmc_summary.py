#!/usr/bin/env python3
"""Per-kernel HBM traffic and matrix-core occupancy of the bench step from rocprofv3 PMC passes (collected in SEPARATE runs, each with
--kernel-trace only: FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

  export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/X/fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/X/write -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/X/sq -- python3 bench.py ...
  python3 tools/pmc_summary.py gpurun_out/X 5 profiles/r02_pmc.json [top]

Corrections (guide, HBM section): both size counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, which
is what the kernels here issue: read bytes = 2 * FETCH_SIZE * 1024, write bytes = WRITE_SIZE * 1024.  MFMA occupancy =
SQ_VALU_MFMA_BUSY_CYCLES / (launch duration * clock * 1024 SIMDs) with the nominal 2.4 GHz clock (the chip clocks lower under load, so
the figure is a lower bound of the per-cycle occupancy)."""
import collections
import csv
import glob
import json
import os
import sys

CLOCK_HZ, SIMDS = 2.4e9, 1024


def counters(d):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return acc
    for r in csv.DictReader(open(fs[0])):
        acc[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        acc[(r["Kernel_Name"], str(g))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    root, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    fetch, write, sq = counters(root + "/fetch"), counters(root + "/write"), counters(root + "/sq")
    dur = durations(root + "/sq")
    rows, step_traffic, step_time, step_mfma = [], 0.0, 0.0, 0.0
    for key, ds in dur.items():
        name, grid = key
        n = len(ds)
        rd = 2 * 1024 * (mean(fetch[key]["FETCH_SIZE"]) or 0.0)
        wr = 1024 * (mean(write[key]["WRITE_SIZE"]) or 0.0)
        t = mean(ds)
        mf = mean(sq[key]["SQ_VALU_MFMA_BUSY_CYCLES"]) or 0.0
        step_traffic += (rd + wr) * n / steps
        step_time += t * n / steps
        step_mfma += mf * n / steps
        rows.append({"kernel": name[:200], "grid": int(grid), "launches_per_step": round(n / steps, 2), "avg_us": round(t * 1e6, 2),
                     "ms_per_step": round(t * n / steps * 1e3, 4), "read_bytes": int(rd), "write_bytes": int(wr), "traffic": int(rd + wr),
                     "hbm_gbs": round((rd + wr) / t * 1e-9, 1), "hbm_frac_of_8tbs": round((rd + wr) / t / 8e12, 4),
                     "mfma_busy_cycles": int(mf), "mfma_occupancy": round(mf / (t * CLOCK_HZ * SIMDS), 4),
                     "sq_wait_inst_any": int(mean(sq[key]["SQ_WAIT_INST_ANY"]) or 0), "sq_wave_cycles": int(mean(sq[key]["SQ_WAVE_CYCLES"]) or 0),
                     "sq_insts_valu": int(mean(sq[key]["SQ_INSTS_VALU"]) or 0)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from source_stamp import git_head, source_stamp

    doc = {"stamp": {"source_sha16": source_stamp(), "git_head": git_head(),
                     "note": "sha256 over iseg_amd/csrc/* and the host files listed in tools/source_stamp.py; bench.py reports these figures only on an identical tree"},
           "source": "rocprofv3 --kernel-trace --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*) over `bench.py --steps 3 --warmup 2`",
           "corrections": "read = 2 * FETCH_SIZE KiB (gfx950 wide coalesced reads report half), write = WRITE_SIZE KiB; MFMA occupancy at the nominal 2.4 GHz",
           "step": {"kernel_time_ms": round(step_time * 1e3, 3), "traffic_bytes": int(step_traffic),
                    "hbm_frac_of_8tbs": round(step_traffic / step_time / 8e12, 4),
                    "mfma_occupancy": round(step_mfma / (step_time * CLOCK_HZ * SIMDS), 4)},
           "kernels": rows[:top]}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["step"]))
    for r in rows[:12]:
        print(f'{r["ms_per_step"]:.3f} ms  {r["avg_us"]:7.1f} us  {r["hbm_gbs"]:7.0f} GB/s  mfma {r["mfma_occupancy"]:.3f}  {r["kernel"][:70]} grid {r["grid"]}')


if __name__ == "__main__":
    main()
