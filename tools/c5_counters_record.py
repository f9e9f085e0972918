#!/usr/bin/env python3
"""Turns PMC passes of the small-set batch (tools/pmc_passes.sh ... --workload c5) into profiles/c5_counters.json, the
record bench.py quotes the batch kernel's vector-ALU issue fraction from.

  python tools/c5_counters_record.py --bench gpurun_out/<c5 bench line>.json gpurun_out/<passes dir> -o profiles/c5_counters.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from traffic_record import counters  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--bench", required=True, help="JSON line of an unprofiled `bench.py --workload c5` run of the same build")
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "profiles", "c5_counters.json"))
    args = ap.parse_args()
    with open(args.bench) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    kernel = line["roofline"]["kernel"]
    got = counters(args.dir, {kernel: kernel})
    c = got.get(kernel, {})
    rec = {"note": "mean per launch of the batch kernel over rocprofv3 --pmc passes of `bench.py --workload c5`",
           "kernel": kernel, "batch_sha": bench.batch_sources_sha(), "ms_per_step": line["ms_per_step"],
           "sets": line["config"]["sets"], "slots": line["config"]["slots"], "source": os.path.relpath(args.dir, ROOT)}
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_I8", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES",
                 "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU"):
        if name in c:
            rec[name] = c[name]
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
