"""gpurun_out/pmcstep_<tag>_{fetch,write,sq,sq2,l2}/ (tools/gpu_pmc_step.sh) -> one text table, one line per kernel:
launches per step, average duration, every counter averaged per launch, derived shares.
    python tools/pmc_step_summary.py <tag> <out.txt>"""
import collections
import csv
import glob
import os
import sys

tag, out_path = sys.argv[1], sys.argv[2]
steps = int(os.environ.get("PMC_STEPS", "3"))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
grid = {}
for p in ("fetch", "write", "sq", "sq2", "l2"):
    for f in glob.glob(f"gpurun_out/pmcstep_{tag}_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[k] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"))
    if p == "fetch":
        for f in glob.glob(f"gpurun_out/pmcstep_{tag}_{p}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = [f"# rocprofv3 --pmc passes (separate runs) over tools/pmc_step.py ({os.environ.get('PMC_CFG', 'cfg2')}, "
         f"B={os.environ.get('PMC_B', 'default')}, {steps} eager steps), MI355X; tools/gpu_pmc_step.sh",
         "# per kernel: launches per step | avg us under the profiler (serialised, not the in-step time) | grid, wg, lds, vgpr | counters averaged per launch",
         "# FETCH_SIZE / WRITE_SIZE in KB (read bytes = 2 x FETCH_SIZE for 16-B/lane reads on gfx950); SQ_* cycle counters are quad-cycles summed over waves",
         "# derived: mfma_share = 4*SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES) where present; waves = SQ_WAVES; wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES; stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"]
tot = sum(sum(v) for v in dur.values()) or 1.0
for k in sorted(cnt, key=lambda k: -sum(dur.get(k, [0]))):
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    n = len(next(iter(cnt[k].values())))
    d = dur.get(k, [0.0])
    extra = []
    if c.get("SQ_WAVE_CYCLES"):
        extra.append(f"wait_share {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}")
        extra.append(f"stall_share {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}")
        extra.append(f"active_share {c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}")
    if c.get("TCC_REQ_sum"):
        extra.append(f"l2_hit {c['TCC_HIT_sum'] / max(1.0, c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.2f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        extra.append(f"lds_conflict_share {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    vals = " ".join(f"{a}={b:.0f}" for a, b in sorted(c.items()))
    lines.append(f"{k[:150]} | per_step {n / steps:.1f} | avg_us {sum(d) / len(d):.2f} | time_share {sum(d) / tot:.3f} | "
                 f"grid,wg,lds,vgpr,sgpr {grid[k]} | {vals} | {' '.join(extra)}")
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(l[:400] for l in lines[:40]))
