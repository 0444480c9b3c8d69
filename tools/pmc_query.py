"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (FETCH_SIZE / WRITE_SIZE in KiB; on this part FETCH_SIZE reports half of the bytes
read: profiles/r02_counter_calibration.json).   python tools/pmc_query.py <results.db> ... [--match substring,substring]"""
import re
import sqlite3
import sys

dbs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--match=")]
match = match[0] if match else ["_kernel"]
rows = {}
for db in dbs:
    for name, cn, avg, n in sqlite3.connect(db).execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
        if any(k in name for k in match):
            m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
            rows.setdefault(m.group(1) if m else name[:60], {})[cn] = avg
for k, v in rows.items():
    extra = ""
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        extra = "  -> %.1f MB read, %.1f MB written" % (2 * v["FETCH_SIZE"] * 1024 / 1e6, v["WRITE_SIZE"] * 1024 / 1e6)
    print("%-50s %s%s" % (k, "  ".join("%s %.0f" % (c, x) for c, x in sorted(v.items())), extra))
