#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short markdown table (for profiles/)."""
import csv, re, sys
from pathlib import Path

def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if "at::native" in name:
        m = re.search(r"(\w+_kernel\w*|\w+Functor\w*|normal_kernel|copy_kernel\w*)", name)
        return "torch:" + (m.group(1) if m else name[:40])
    return name[:60]

def main(d, out, note=""):
    d = Path(d)
    agg = {}
    stats = next(d.rglob("*_kernel_stats.csv"), None)
    if stats is not None:
        for r in csv.DictReader(open(stats)):
            k = short(r["Name"])
            a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
            a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    else:                                   # rocprofv3's default output: a rocpd SQLite database (view `kernels`)
        import sqlite3
        db = sqlite3.connect(str(next(d.rglob("*_results.db"))))
        for name, n, tot, mn, mx in db.execute("select name, count(*), sum(duration), min(duration), max(duration) from kernels group by name"):
            k = short(name)
            a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
            a[0] += n; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    tot = sum(a[1] for a in agg.values())
    lines = [f"# rocprofv3 --kernel-trace --stats summary\n", note, "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"| `{k}` | {a[0]} | {a[1]/1e6:.2f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/tot:.2f} |")
    lines.append(f"\nTotal kernel time {tot/1e6:.1f} ms over {sum(a[0] for a in agg.values())} dispatches.")
    Path(out).write_text("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
