"""Trim a rocprofv3 --kernel-trace --stats `*_kernel_stats.csv` to the rows that matter (template names
shortened) so that the summary can be committed.  usage: python profiles/summarize.py <kernel_stats.csv> <out.csv> [N]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) < 120 else name[:117] + "..."


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:n + 1]:
            w.writerow([short(r[0])] + r[1:])


if __name__ == "__main__":
    main()
