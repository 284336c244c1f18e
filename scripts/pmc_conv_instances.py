#!/usr/bin/env python
"""
HBM traffic per launch of EVERY instance of the decoder convolution as the timed tree runs it (round 4; round 5: the
Winograd kernel, `python scripts/pmc_conv_instances.py --wino ...`):

    conv5x5_dec_f16x3_kernel<1, false, false>   layer 1: collapsed layer-0 input, operand planes out
    conv5x5_dec_f16x3_kernel<0, true, false>    layer 2: operand planes in and out
    conv5x5_dec_f16x3_kernel<0, true, true>     layer 3: operand planes in, the folded tail's 36 tap planes out
    conv5x5_wino_f16x3_kernel<1, false> / <0, false> / <0, true>   the same three layers (fp32 x 16 hand-over)

from two rocprofv3 PMC passes over `python3 scripts/decode_only.py 68 3` (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, the
program directly behind `--`; corrections of /opt/skills/guides/MI355X_MICROARCH.md, section HBM:
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- KiB units, gfx950 counts half of a wide coalesced read).

    python scripts/pmc_conv_instances.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Algorithmic bytes per 64 x 64 slot image (fp32-sized operands: two fp16 planes = 4 bytes per value):
    layer 1: 25 x 64 x 4 B of tap sums in (+ the shared 1 MiB position table, once per launch), 64 ch x 4096 px x 4 B out
    layer 2: 1 MiB in, 1 MiB out           layer 3: 1 MiB in, 36 x 4096 x 4 B out
"""
import csv
import json
import sys

MIB = 1024.0 * 1024.0
INSTANCES = {
    "<1, false, false>": {"layer": 1, "alg_in": 25 * 64 * 4.0, "alg_out": MIB, "shared_in": MIB},
    "<0, true, false>": {"layer": 2, "alg_in": MIB, "alg_out": MIB, "shared_in": 0.0},
    "<0, true, true>": {"layer": 3, "alg_in": MIB, "alg_out": 36 * 4096 * 4.0, "shared_in": 0.0},
}


WINO_INSTANCES = {
    "<1, false>": {"layer": 1, "alg_in": 25 * 64 * 4.0, "alg_out": MIB, "shared_in": MIB},
    "<0, false>": {"layer": 2, "alg_in": MIB, "alg_out": MIB, "shared_in": 0.0},
    "<0, true>": {"layer": 3, "alg_in": MIB, "alg_out": 36 * 4096 * 4.0, "shared_in": 0.0},
}
KERNEL = "conv5x5_dec_f16x3_kernel"
TILES_PER_IMAGE = 8.0                                               # 8 x 64 tiles of a 64 x 64 slot image


def per_instance(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or KERNEL not in r["Kernel_Name"]:
            continue
        for key in INSTANCES:
            if KERNEL + key in r["Kernel_Name"].replace("kernel <", "kernel<"):
                out.setdefault(key, []).append((int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    return out


def main():
    global INSTANCES, KERNEL, TILES_PER_IMAGE
    if sys.argv[1] == "--wino":
        INSTANCES, KERNEL, TILES_PER_IMAGE = WINO_INSTANCES, "conv5x5_wino_f16x3_kernel", 16.0     # 4 x 64 tiles
        sys.argv.pop(1)
    fpath, wpath, out = sys.argv[1:4]
    fetch, write = per_instance(fpath, "FETCH_SIZE"), per_instance(wpath, "WRITE_SIZE")
    rec = {"kernel": KERNEL + (" (f16x3-wino)" if "wino" in KERNEL else " (f16x3)") + ", the three instances of the default decoder",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 1/2 of wide "
                         "coalesced reads; KiB units); separate --pmc passes", "instances": {}}
    tot_hbm = tot_alg = 0.0
    for key, spec in INSTANCES.items():
        f, w = fetch.get(key, []), write.get(key, [])
        if not f or not w:
            continue
        gmax = max(g for g, _, _ in f)                              # full chunks only
        fv = [v for g, _, v in f if g == gmax]
        wv = [v for g, _, v in w if g == gmax]
        wgs = gmax // f[0][1]
        images = wgs / TILES_PER_IMAGE
        hbm = (2.0 * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024.0
        alg = images * (spec["alg_in"] + spec["alg_out"]) + spec["shared_in"]
        rec["instances"][key] = {
            "layer": spec["layer"], "launches_fetch_pass": len(fv), "launches_write_pass": len(wv),
            "fetch_size_kib_mean_raw": sum(fv) / len(fv), "write_size_kib_mean": sum(wv) / len(wv),
            "slot_images_per_launch": images, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
            "hbm_over_algorithmic": hbm / alg}
        tot_hbm += hbm
        tot_alg += alg
    n = len(rec["instances"])
    if n:
        imgs = next(iter(rec["instances"].values()))["slot_images_per_launch"]
        rec.update({"slot_images_per_launch": imgs, "hbm_bytes_per_launch": tot_hbm / n,
                    "algorithmic_bytes_per_launch": tot_alg / n, "hbm_over_algorithmic": tot_hbm / tot_alg,
                    "note": "hbm_bytes_per_launch = mean over the three instances, each launched once per chunk "
                            "(the launch-weighted average of the timed tree)"})
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
