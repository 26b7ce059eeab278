"""Where the wall time of a two-stream step goes: from a rocprofv3 kernel trace, per step (the last one in the trace)
the union of kernel intervals, split into time with an MFMA kernel running, time with only other kernels, and idle gaps."""
import csv
import json
import os
import re
import sys

# usage: timeline.py <kernel_trace.csv> [out.json]   (the JSON distillate is what profiles/<tag>_timeline.json holds)
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by the Adam launches (two per step, the generator's is the last kernel of a step)
adam = [i for i, e in enumerate(ev) if "adam_clip" in e[2]]
assert len(adam) >= 4, "need at least two steps"
lo, hi = adam[-3] + 1, adam[-1] + 1            # the last complete step: after the previous step's second Adam
step = ev[lo:hi]
t0, t1 = step[0][0], max(e[1] for e in step)
is_mfma = lambda n: bool(re.search(r"tapgemm|wgrad_(halo|kernel|bf16)", n))
pts = []
for s, e, n in step:
    k = 1 if is_mfma(n) else 0
    pts.append((s, 0, k))
    pts.append((e, 1, k))
pts.sort()
cnt = [0, 0]
last = t0
acc = {"mfma": 0, "other_only": 0, "idle": 0}
by_other = {}
open_other = {}
for t, typ, k in pts:
    dt = t - last
    if cnt[1] > 0:
        acc["mfma"] += dt
    elif cnt[0] > 0:
        acc["other_only"] += dt
    else:
        acc["idle"] += dt
    last = t
    cnt[k] += 1 if typ == 0 else -1
wall = (t1 - t0) / 1e6
print(f"step wall {wall:.3f} ms  kernels {len(step)}")
for k, v in acc.items():
    print(f"  {k:11s} {v / 1e6:8.3f} ms")
# exposed time by non-MFMA kernel name: portions of each non-MFMA kernel during which no MFMA kernel runs
mf = sorted((s, e) for s, e, n in step if is_mfma(n))
merged = []
for s, e in mf:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
import bisect
starts = [m[0] for m in merged]
def uncovered(s, e):
    tot = e - s
    i = max(0, bisect.bisect_right(starts, s) - 1)
    while i < len(merged) and merged[i][0] < e:
        a, b = max(s, merged[i][0]), min(e, merged[i][1])
        if b > a:
            tot -= b - a
        i += 1
    return tot
exp = {}
for s, e, n in step:
    if not is_mfma(n):
        nm = re.sub(r"[(<].*", "", n)[:40]
        exp[nm] = exp.get(nm, 0) + uncovered(s, e)
print("exposed (no MFMA kernel running) by kernel:")
for nm, v in sorted(exp.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {nm:42s} {v / 1e6:7.3f} ms")

if len(sys.argv) > 2:
    json.dump({"_meta": {"tree": os.environ.get("SHM_TREE_SHA", "unknown"), "csrc_sha16": os.environ.get("SHM_CSRC_SHA", "unknown"),
                         "what": "rocprofv3 kernel trace of the PRODUCTION two-stream step (last complete step of the trace), tools/timeline.py"},
               "wall_ms": round(wall, 3), "mfma_running_ms": round(acc["mfma"] / 1e6, 3), "elementwise_only_ms": round(acc["other_only"] / 1e6, 3),
               "idle_ms": round(acc["idle"] / 1e6, 3), "kernels": len(step),
               "exposed_ms_by_kernel": {nm: round(v / 1e6, 3) for nm, v in sorted(exp.items(), key=lambda kv: -kv[1])[:20] if v > 0}},
              open(sys.argv[2], "w"), indent=1)
