"""Dumps the per-kernel summary (top_kernels view) of a rocprofv3 results .db as markdown -- what gets committed
under profiles/."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"| `{name[:110]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
try:
    rows = list(c.execute("select name, counter_name, sum(value), count(*) from counters_collection group by name, counter_name"))
    if rows:
        print("\n| kernel | counter | sum | dispatches |\n|---|---|---|---|")
        for r in rows:
            print(f"| `{r[0][:90]}` | {r[1]} | {r[2]:.6g} | {r[3]} |")
except sqlite3.Error as e:
    print("(no counters:", e, ")")
