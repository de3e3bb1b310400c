import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
cur = c.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
rows = list(c.execute("select name, start, end from kernels order by start"))
# last proof: take the last 40 kernels
rows = rows[-48:]
t0 = rows[0][1]
prev_end = None
for name, s, e in rows:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name.split('(')[0][:70]}")
    prev_end = e
