#!/usr/bin/env python3
"""Per-launch HBM-side traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes:
FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Corrections per that
guide's HBM section: counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled.
usage: pmc_summary.py <fetch_dir> <write_dir> <kernel-substring> <out.json> [note]"""
import csv, hashlib, json, sys
from pathlib import Path


def gemm_src_sha16() -> str:
    """Hash of the sources the measured GEMM kernels are built from (the same function lives in bench.py, which refuses a
    summary whose hash is not that of the tree it runs from: the box has no git history to compare commits with)."""
    csrc = Path(__file__).resolve().parents[1] / "licv-vqa_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted([f for f in csrc.glob("gemm*") if f.is_file()] + [csrc / "common.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def total(d, counter, sub):
    n, s = 0, 0.0
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
                n += 1
                s += float(r["Counter_Value"])
    return n, s


def main(fd, wd, sub, out, note=""):
    nf, f = total(fd, "FETCH_SIZE", sub)
    nw, w = total(wd, "WRITE_SIZE", sub)
    res = {"kernel": sub, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_bytes_per_launch": 2.0 * 1024.0 * f / max(nf, 1), "write_bytes_per_launch": 1024.0 * w / max(nw, 1),
           "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B)", "note": note}
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["gemm_src_sha16"] = gemm_src_sha16()
    Path(out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
