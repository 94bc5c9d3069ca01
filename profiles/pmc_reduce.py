"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE dumps (profiles/collect.sh) to HBM-side bytes per launch.

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and FETCH_SIZE reads half on gfx950
(MI355X_MICROARCH.md, HBM/rocprofv3 section); the factor is re-checked on every run with the 1 GiB k_copy
launches of bench.py's bandwidth probe (expected 2 GiB of traffic)."""
import collections
import csv
import glob
import json
import sys

NAMES = [("k_tend", "k_tend"), ("k_rfft3_unpack", "k_rfft3_unpack"), ("k_thomas_corr", "k_thomas_corr"), ("k_oml_step", "k_oml_step"), ("k_oml_entoc", "k_oml_entoc"), ("k_dst64_unpack", "k_dst_inv"), ("k_dst64<", "k_dst_fwd"), ("k_dst_box", "k_dst_fwd"),
         ("k_rfft_cyc<true", "k_dst_inv"), ("k_rfft_cyc<false", "k_dst_fwd"),
         ("k_thomas", "k_thomas"), ("k_unpack", "k_unpack"), ("k_constr", "k_constr"), ("k_lf_average", "k_lf_average"),
         ("k_copy", "k_copy_1GiB_calibration")]


def short(name):
    for pat, s in NAMES:
        if pat in name:
            return s
    return None


def per_launch(out, counter):
    f = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (out, counter))[0]
    tot, n = collections.defaultdict(float), collections.Counter()
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        per_dispatch[(k, r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, _), v in per_dispatch.items():
        tot[k] += v
        n[k] += 1
    return {k: tot[k] / n[k] for k in tot}, dict(n)


def main():
    out = sys.argv[1]
    fetch, nf = per_launch(out, "FETCH_SIZE")
    write, _ = per_launch(out, "WRITE_SIZE")
    res = {k: int(round((2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0)) for k in fetch}
    res["_raw_KiB"] = {k: {"FETCH_SIZE": round(fetch[k], 1), "WRITE_SIZE": round(write.get(k, 0.0), 1), "launches": nf[k]}
                       for k in fetch}
    res["_note"] = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                    "in separate passes (profiles/collect.sh); the factor 2 is the gfx950 FETCH_SIZE correction of "
                    "MI355X_MICROARCH.md, checked on the 1 GiB k_copy launches (expect 2147483648).")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
