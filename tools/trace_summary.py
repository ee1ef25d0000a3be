"""Summarise a rocprofv3 kernel trace: per-kernel time per step, image / text / common chain totals, wall time."""
import collections, csv, sys
f, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: drop the first third
rows = rows[len(rows) // 3:]
img = ("conv", "bce", "sigmoid_clamp")
txt = ("attn", "ln_", "embed", "ce_time", "dropout", "head_bcast", "permute_mask", "time_sum", "time_bcast")
agg = collections.defaultdict(list)
chain = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = n[:36] + " " + str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"]
    agg[key].append(d)
    c = "img" if any(t in n for t in img) else "txt" if any(t in n for t in txt) else "gemm" if "gemm" in n else "other"
    chain[c] += d
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
frac = len(rows) / (len(rows) * 1.5)
nst = steps * 2 / 3
print("wall us/step ~", wall / nst / 1e3, " launches/step ~", len(rows) / nst)
for c, v in chain.items():
    print(f"  {c:6s} {v / nst / 1e3:8.1f} us/step")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"{sum(v) / nst / 1e3:7.1f}us/step n={len(v) / nst:5.1f} avg={sum(v) / len(v) / 1e3:7.1f}us  {k}")
# overlap: fraction of wall time with >= 2 kernels running
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = ev[0][0]; t = collections.Counter()
for ts, dlt in ev:
    t[min(cur, 3)] += ts - last; last = ts; cur += dlt
tot = sum(t.values())
print("concurrency: " + ", ".join(f"{k} running: {100 * v / tot:.1f}%" for k, v in sorted(t.items())))
