"""Turn rocprofv3 CSV output (gpurun_out/<dir>) into the text summaries committed under profiles/.
   python tools/summarize_prof.py stats <dir> <title...>      kernel_stats.csv -> table
   python tools/summarize_prof.py pmc <dir> <counter> ...     counter_collection.csv -> per-kernel per-launch averages"""
import collections, csv, glob, sys


def stats(d):
    f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':92s}{'calls':>6s}{'avg_us':>12s}{'pct':>8s}")
    for r in rows[:18]:
        print(f'{r["Name"][:90]:92s}{int(r["Calls"]):6d}{float(r["AverageNs"]) / 1e3:12.1f}{100 * float(r["TotalDurationNs"]) / tot:8.2f}')


def pmc(d):
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            a = acc[(r["Counter_Name"], r["Kernel_Name"][:70])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for (c, k), (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0] / kv[1][1]):
            if v / n > 1e3:
                print(f"   {c:12s} {k:72s} n={n:4d} avg={v / n:.4g}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
