#!/usr/bin/env python3
"""profiles/kernel_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output):
HBM-side bytes per launch of each C-ABI launch unit bench.py reports a roofline for = sum over the kernels the unit
enqueues of (2 x FETCH_SIZE + WRITE_SIZE) KiB (FETCH doubled: the gfx950 correction for wide coalesced reads,
MI355X_MICROARCH.md; WRITE_SIZE uncorrected).

    python tools/make_traffic_json.py FETCH_DIR WRITE_DIR OUT.json [SUMMARY.txt]
"""
import collections
import csv
import glob
import json
import re
import sys

UNITS = {   # ABI unit -> [(kernel substring, launches of it per unit call)]
    "mcl_dense_bn1_bwd": [("bn1_bwd_kernel<0", 1), ("bn1_bwd_finalize_kernel", 1), ("bn1_bwd_kernel<1", 1)],
    "mcl_dense_conv3x3_bwd": [("conv3x3_bwd_", 1), ("bn2_dz_kernel", 1)],          # flat + row-walking forms (launch-weighted)
    "mcl_dense_conv1x1_fwd": [("conv1x1_fwd_kernel", 1)],
    "mcl_dense_conv3x3_fwd": [("conv3x3_fwd_", 1)],                               # flat + row-walking forms
    "mcl_conv1x1_wrw_det": [("wrw_partial_kernel<2", 1), ("wrw_partial_kernel<0", 1)],
    "mcl_dense_bn1_wrw": [("wrw_partial_kernel<1", 1)],
    "mcl_dense_bn1_dx": [("bn1_bwd_kernel<1", 1)],
    "mcl_dense_conv3x3_wrw_det": [("conv3x3_wrw_k", 1)],
    "mcl_adam_table_step_dev": [("adam_table_kernel", 1)],
    "mcl_adam_step_dev": [("adam_kernel(", 1)],
    "mcl_adam_step_dev_shadow": [("adam_kernel<true>", 1)],
}


def load(d):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void\s+", "", name)
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    per_kernel = {}
    names = sorted({k[0] for k in list(fetch) + list(write)})
    lines = []
    for n in names:
        f = fetch.get((n, "FETCH_SIZE"), [])
        w = write.get((n, "WRITE_SIZE"), [])
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        per_kernel[n] = (2.0 * fk * 1024.0, wk * 1024.0, max(len(f), len(w)))
        lines.append(f"{n.split('(')[0][-70:]:72s} launches {max(len(f), len(w)):5d}  FETCH_SIZE {fk:12.0f} KiB (x2 = "
                     f"{2 * fk * 1024 / 1e6:9.2f} MB)  WRITE_SIZE {wk:12.0f} KiB ({wk * 1024 / 1e6:9.2f} MB)")
    out = {"_method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --no_cpu_baseline "
                      "--steps 3 --warmup 1 --profile_steps 0`; per-launch averages; KiB -> bytes; FETCH_SIZE doubled "
                      "(gfx950 correction for wide coalesced reads), WRITE_SIZE uncorrected; per launch unit = sum over "
                      "the kernels it enqueues"}
    for unit, parts in UNITS.items():
        tot, found = 0.0, False
        for sub, mult in parts:
            ks = [k for k in per_kernel if sub in k]
            if not ks:
                continue
            found = True
            # launch-weighted mean over template instances of this kernel
            n_l = sum(per_kernel[k][2] for k in ks)
            tot += mult * sum((per_kernel[k][0] + per_kernel[k][1]) * per_kernel[k][2] for k in ks) / max(n_l, 1)
        if found:
            out[unit] = round(tot)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write("\n".join(lines) + "\n")
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
