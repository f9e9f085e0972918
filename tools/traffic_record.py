#!/usr/bin/env python3
"""Turns the PMC passes of tools/pmc_passes.sh into profiles/traffic.json, the record bench.py quotes
`roofline.traffic` from.

  python tools/traffic_record.py --bench gpurun_out/<bench line>.json --workload c3 \
         gpurun_out/<f64 passes dir> gpurun_out/<fr passes dir> -o profiles/traffic.json

Per kernel: fabric bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE is in KiB and
counts 64 bytes per read request, while every request these kernels make is 128 bytes: TCC_EA0_RDREQ_128B equals
TCC_EA0_RDREQ in the same passes; WRITE_SIZE is in KiB).  The record carries the hash of the kernel sources and
the launch time of an unprofiled run, so that bench.py can tell whether it still describes the kernel it runs.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_sha)

NAMES = {"k_step_f64_blocked": "k_step_f64_blocked<u8>", "k_step_fr_blocked": "k_step_fr_blocked", "k_step_fr_raw": "k_step_fr_raw"}


def counters(root, names=None):
    names = names or NAMES
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Kernel_Name"], row["Counter_Name"])
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1] += 1
    out = defaultdict(dict)
    for (kernel, ctr), (tot, cnt) in acc.items():
        for frag, name in names.items():
            if frag in kernel:
                out[name][ctr] = tot / cnt
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--bench", required=True, help="JSON line of an unprofiled bench.py run of the same build")
    ap.add_argument("--workload", default="c3")
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    args = ap.parse_args()
    with open(args.bench) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    launch = {line["roofline"]["kernel"]: line["roofline"]["launch_us"]}
    if "fr_exact" in line:
        launch[line["fr_exact"]["roofline"]["kernel"]] = line["fr_exact"]["roofline"]["launch_us"]
    rec = {"note": "fabric bytes per iteration launch from rocprofv3 --pmc passes (tools/pmc_passes.sh) of bench.py's workload; "
                   "traffic_bytes = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024",
           "workload": args.workload, "csrc_sha": bench.kernel_sources_sha(), "kernels": {}}
    for d in args.dirs:
        for name, c in counters(d).items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or name not in launch:
                continue
            k = {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                 "traffic_bytes": int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
                 "launch_us": launch[name], "source": os.path.relpath(d, ROOT)}
            for extra in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum",
                          "TCP_TCC_READ_REQ_sum", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"):
                if extra in c:
                    k[extra] = c[extra]
            rec["kernels"][name] = k
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
