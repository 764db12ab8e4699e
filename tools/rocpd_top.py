"""Print the top kernels of a rocprofv3 run (its rocpd sqlite database) as CSV: name, calls, total us, average us, percent.
usage: python tools/rocpd_top.py <results.db> [n=40] [> profiles/rNN_x_kernel_stats.csv]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("name,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in rows[:n]:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short) if len(short) > 120 else short
    short = short.replace(",", ";")
    print(f"{short[:110]},{calls},{total:.1f},{avg:.2f},{pct:.2f}")
