#!/usr/bin/env python3
"""Print a compact table from a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv)."""
import csv
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:34]


def main(path, top=14):
    rows = list(csv.DictReader(open(path)))
    print("%-34s %6s %10s %12s %7s" % ("kernel", "calls", "avg_us", "total_us", "pct"))
    for r in rows[:top]:
        print("%-34s %6s %10.1f %12.1f %7.2f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                 float(r["TotalDurationNs"]) / 1e3, float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
