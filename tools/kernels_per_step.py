"""Kernels per step from a rocprofv3 --kernel-trace --stats run: python tools/kernels_per_step.py <dir> <steps incl. warm-up>
Lists the heaviest kernels as calls / step and ms / step, and sums the framework's own small kernels (at::native, rocclr)."""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:96]:98s}{int(r['Calls']) / n:8.1f}/step {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / n / 1e6:7.3f} ms/step")
small = [r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print(f"framework kernels: {sum(int(r['Calls']) for r in small) / n:.0f} calls/step, {sum(float(r['TotalDurationNs']) for r in small) / n / 1e6:.3f} ms/step")
