"""alone (one stream) vs in-step (two streams) per-kernel averages from two rocprofv3 kernel_stats.csv files"""
import csv, sys
a, b, steps = sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 36.0
def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3) for r in csv.DictReader(open(f))}
A, B = load(a), load(b)
ta = tb = 0
rows = []
for n in A:
    if n in B:
        ca, aa, sa = A[n]; cb, ab, sb = B[n]
        rows.append((sb / steps, n, ca / steps, aa, ab))
        ta += sa / steps; tb += sb / steps
rows.sort(reverse=True)
print(f"sum of kernel time per step: alone {ta:.0f} us, in step {tb:.0f} us")
for s, n, c, aa, ab in rows[:40]:
    print(f"{n[:84]:84s} x{c:4.1f}  alone {aa:7.1f}  in-step {ab:7.1f}  ({ab / aa:4.2f}x)  {s:7.1f} us/step")
