"""Two rocprofv3 `--kernel-trace --stats` runs of the same command (two builds, one box, one call): per kernel name, calls and total time in
both, sorted by the difference.  usage: python tools/ab_kernel_stats.py <dir_old> <dir_new>"""
import csv, glob, os, re, sys


def load(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"])
        name = re.sub(r"PDF16bl$|iS2_S2_$", "", name)[:70]          # (parameter lists that changed between the builds)
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]) / 1e3)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
print(f"{'kernel':72s} {'calls':>6s} {'us old':>10s} {'calls':>6s} {'us new':>10s} {'diff':>9s}")
for d, k, ca, ta, cb, tb in sorted(rows):
    if abs(d) >= 20 or ca != cb:
        print(f"{k:72s} {ca:6d} {ta:10.0f} {cb:6d} {tb:10.0f} {d:+9.0f}")
print(f"{'total':72s} {sum(v[0] for v in a.values()):6d} {sum(v[1] for v in a.values()):10.0f} {sum(v[0] for v in b.values()):6d} {sum(v[1] for v in b.values()):10.0f}")
