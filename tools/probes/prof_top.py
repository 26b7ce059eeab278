"""Print the per-step kernel table of a rocprofv3 results .db (steps = launches of adam_clip / 2)."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/*/*.db"))[-1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average from top_kernels"))
steps = max(1, [r[1] for r in rows if "adam_clip" in r[0]][0] // 2)
tot = sum(r[2] for r in rows)
print(f"{steps} steps, {tot / steps / 1e3:.2f} ms of kernel time per step")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[0][:100]:100s} {r[1] // steps:4d}/step {r[2] / steps / 1e3:8.3f} ms  avg {r[3]:8.1f} us")
