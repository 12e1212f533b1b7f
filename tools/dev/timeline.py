"""dev: per-dispatch timeline (start, duration, gap to the previous end) of the last N dispatches of a rocpd database"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print(tabs); sys.exit(0)
cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
rows = list(c.execute("select name, start, end, stream_id, queue_id from %s order by start" % view)) if "stream_id" in cols else \
       [r + (0, 0) for r in c.execute("select name, start, end from %s order by start" % view)]
rows = rows[-n:]
t0 = rows[0][1]
last_end = rows[0][1]
for name, s, e, st, q in rows:
    short = name.split("(")[0].split("::")[-1][:34]
    print("%9.1f us  +%8.1f us  gap %8.1f  s%-3s q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, st, q, short))
    last_end = max(last_end, e)
