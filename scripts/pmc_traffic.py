#!/usr/bin/env python
"""
HBM traffic per launch of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE are collected in SEPARATE runs: they do not fit the TCC slots together).

Unit and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM) and
cdna_hip_programming.md section 7:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
  - both counters are in KiB;
  - on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced
    streaming read -> doubled (the conv kernel stages its tiles with 16-byte loads).  Calibrated for
    the f16f8 kernel's pattern (one 64-byte quarter of every 256-byte pixel per pass) with
    scripts/probes/fetch_calib.hip: FETCH_SIZE = payload and the time of 2x the payload, i.e. whole
    128-byte lines move and the x2 rule holds there too;
  - WRITE_SIZE is exact for 16-byte-per-lane streaming stores; the conv epilogue stores 4 B per
    lane in 128-byte segments, so the write side is an uncalibrated (but plausible) figure.

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
           <kernel substring> <out.json> [workgroups per slot image: 16 (8x32 tiles) | 8 (8x64 tiles)]
"""
import csv
import json
import sys


def per_kernel(path, counter, needle):
    """ counter values of the launches with the LARGEST grid (full chunks), and that grid """
    recs = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            recs.append((int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    gmax = max(g for g, _, _ in recs)
    vals = [v for g, _, v in recs if g == gmax]
    return vals, (gmax, recs[0][1])


def main():
    fpath, wpath, needle, out = sys.argv[1:5]
    wg_per_image = float(sys.argv[5]) if len(sys.argv) > 5 else 16.0
    f, grid = per_kernel(fpath, "FETCH_SIZE", needle)
    w, _ = per_kernel(wpath, "WRITE_SIZE", needle)
    fetch_kib = sum(f) / len(f)
    write_kib = sum(w) / len(w)
    workgroups = grid[0] // grid[1]
    rec = {
        "kernel": needle, "launches_fetch_pass": len(f), "launches_write_pass": len(w),
        "fetch_size_kib_mean_raw": fetch_kib, "write_size_kib_mean": write_kib,
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
        "workgroups_per_launch": workgroups, "slot_images_per_launch": workgroups / wg_per_image,
        "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts "
                      "1/2 of wide coalesced reads; KiB units)",
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
