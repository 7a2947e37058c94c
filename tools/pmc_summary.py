"""Summaries of the rocprofv3 --pmc passes of tools/measure.sh pmc (one counter per pass over tools/pmc_driver.py).
  python tools/pmc_summary.py counter <counter_collection.csv> <driver log> <COUNTER> <out dir>   -> <out dir>/<COUNTER>.txt
  python tools/pmc_summary.py traffic <out dir>                                                  -> <out dir>/traffic.json
  python tools/pmc_summary.py mfma <counter_collection.csv> <out dir>                            -> <out dir>/mfma.json
traffic.json / mfma.json carry the SHA-256 of the kernel sources they were taken on (bench.kernel_source_fingerprint): bench.py reports
them only for the sources being run. FETCH_SIZE is corrected per kernel by the factors calibrated on known byte counts
(profiles/r05_pmc_fetch_write_calibration.txt: the counter reports half of a coalesced streaming read of any width, all of a random gather)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = (("k_prune_march", "ml", "enc", "encoded sample"), ("k_encode4d_fwd", "fl", "n1", "rendered sample"),
           ("k_scatter_emit", "bl", "n1", "rendered sample"), ("k_scatter_accumulate", "bl", "n1", "rendered sample"),
           ("k_encode4d_bwd_tables_lm", "bl", "n1", "rendered sample"), ("k_encode4d_bwd_vectors", "bl", "n1", "rendered sample"))
FETCH_CORRECTION = {"k_scatter_accumulate": 2.0, "k_scatter_emit": 2.0, "k_encode4d_bwd_vectors": 2.0}   # coalesced streams: x 2


def counter(csv_path, log_path, name, out_dir):
    log = open(log_path).read()
    m = re.search(r"PMC_WINDOW steps (\d+) segments (\[.*?\]) march_launches (\d+) encoded (\d+) fwd_launches (\d+) bwd_launches (\d+) "
                  r"rendered (\d+) rays (\d+)", log)
    if not m:
        print("no PMC_WINDOW line", log[-400:])
        raise SystemExit(1)
    steps, segs = m.group(1), m.group(2)
    w = dict(zip(("ml", "enc", "fl", "bl", "n1", "rays"), (int(x) for x in m.groups()[2:])))
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(csv_path)):
        if r["Counter_Name"] == name:
            by[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    unit = 1024.0 if name.endswith("_SIZE") else 1.0   # FETCH_SIZE / WRITE_SIZE count kilobytes
    out = ["# %s, separate pass, window of %s steps, segments %s, %d rays, %d rendered samples" % (name, steps, segs, w["rays"], w["n1"])]
    for kname, lk, uk, what in KERNELS:
        key = [k for k in by if kname in k]
        if not key:
            continue
        launches, units = w[lk], w[uk]
        tot = sum(by[key[0]][-launches:]) * unit
        out.append("%-28s launches %3d  total %.6g  units %d  per %s %.2f  per launch %.6g"
                   % (kname, launches, tot, units, what, tot / max(units, 1), tot / max(launches, 1)))
    open(os.path.join(out_dir, name + ".txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


def traffic(out_dir):
    import bench

    def per(cname, kernel):
        try:
            for line in open(os.path.join(out_dir, cname + ".txt")):
                if line.startswith(kernel + " "):
                    return float(re.search(r"per (?:encoded|rendered) sample ([0-9.eE+-]+)", line).group(1))
        except FileNotFoundError:
            pass
        return None

    def both(k):
        f, w = per("FETCH_SIZE", k), per("WRITE_SIZE", k)
        if f is None or w is None:
            return None
        c = FETCH_CORRECTION.get(k, 1.0)
        return {"fetch_bytes_per_encoded_sample": f * c, "write_bytes_per_encoded_sample": w, "fetch_size_as_reported": f, "fetch_correction": c}

    j = {"kernel_sources_sha256": bench.kernel_source_fingerprint(),
         "source": "tools/measure.sh pmc: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC_sum, one counter per pass, "
                   "8-step window of the default bench configuration after PM_WARM training steps; FETCH_SIZE corrected per kernel "
                   "(x 2 for the kernels whose reads are coalesced streams, x 1 for the gather kernels: "
                   "profiles/r05_pmc_fetch_write_calibration.txt)",
         "header": open(os.path.join(out_dir, "FETCH_SIZE.txt")).readline().strip()}
    for k in ("k_prune_march", "k_encode4d_fwd"):
        b = both(k)
        if b:
            j[k] = b
    e, a = both("k_scatter_emit"), both("k_scatter_accumulate")
    if e and a:
        j["table_scatter"] = {"fetch_bytes_per_encoded_sample": e["fetch_bytes_per_encoded_sample"] + a["fetch_bytes_per_encoded_sample"],
                              "write_bytes_per_encoded_sample": e["write_bytes_per_encoded_sample"] + a["write_bytes_per_encoded_sample"],
                              "l2_atomic_requests_per_sample": (per("TCC_ATOMIC_sum", "k_scatter_emit") or 0.0)
                                                               + (per("TCC_ATOMIC_sum", "k_scatter_accumulate") or 0.0),
                              "kernels": {"k_scatter_emit": e, "k_scatter_accumulate": a}}
    v = both("k_encode4d_bwd_vectors")
    if v:
        v["l2_atomic_requests_per_sample"] = per("TCC_ATOMIC_sum", "k_encode4d_bwd_vectors")
        j["k_encode4d_bwd_vectors"] = v
    json.dump(j, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
    print(json.dumps(j, indent=1))


def mfma(csv_path, out_dir):
    import bench
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(csv_path)):
        if r["Counter_Name"] == "MfmaUtil":
            by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    j = {"kernel_sources_sha256": bench.kernel_source_fingerprint(),
         "source": "tools/measure.sh pmc: rocprofv3 --kernel-trace --pmc MfmaUtil (its own pass) over tools/pmc_driver.py, the default bench "
                   "configuration after PM_WARM training steps; mean of the last 8 launches of each kernel, per cent"}
    for key in ("k_mlp_bwd", "k_color_fwd", "k_density_fwd", "k_prune_march"):
        for name, v in by.items():
            if key in name:
                tail = v[-8:]
                j[key] = round(sum(tail) / len(tail), 3)
                print("%-16s MfmaUtil n=%d mean_last8 %.3f %%" % (key, len(v), j[key]))
    json.dump(j, open(os.path.join(out_dir, "mfma.json"), "w"), indent=1)


if __name__ == "__main__":
    {"counter": counter, "traffic": traffic, "mfma": mfma}[sys.argv[1]](*sys.argv[2:])
