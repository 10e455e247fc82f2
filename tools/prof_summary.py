"""Condense rocprofv3 csv output (kernel trace stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("== kernel trace stats (", root, ")")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:70]
            print(f"{name:70s} calls {row.get('Calls'):>6s} avg_ns {row.get('AverageNs'):>12s} "
                  f"total_ns {row.get('TotalDurationNs'):>12s} pct {row.get('Percentage')}")
print("== PMC (per-dispatch average by kernel)")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in find("*counter_collection.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")[:60]
            c = row.get("Counter_Name")
            v = float(row.get("Counter_Value", 0))
            a = acc[k][c]
            a[0] += v
            a[1] += 1
for k in acc:
    if not any(t in k for t in ("welch", "k_y", "k_x", "csm", "fir", "stft", "deconv", "big", "blue")):
        continue
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print(f"    {c:28s} {s / n:18.1f}  (n={n})")

# which machine code the counters above were taken on: bench.py compares these with the library it runs
# (roofline.traffic_kernel_current) and drops the traffic figure of a kernel that has changed since
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from dsptoolbox_amd import _build
    fps = _build.demangled_fingerprints()
except Exception as ex:  # noqa: BLE001
    fps = {}
    print("== kernel code: not available (", repr(ex)[:100], ")")
if fps:
    print("== compiler of the profiled library:", _build.compiler_id())
    print("== kernel code (sha256[:16] of the gfx950 machine code of the kernels above: dsptoolbox_amd._build.kernel_fingerprints)")
    for k in acc:
        hit = sorted(h for d, h in fps.items() if d[:60] == k)
        if len(hit) == 1:
            print(f"    {hit[0]}  {k}")
