"""Per-kernel average durations of N proofs of the 10^7-gate circuit from a rocprofv3 kernel trace (results.db): the kernels whose
name contains one of the given substrings.  usage: python tools/kern_times.py <results.db> name [name ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in sys.argv[2:]:
    rows = list(c.execute("select name, count(*), avg(end - start), min(end - start) from kernels where name like ? group by name", ("%" + pat + "%",)))
    for name, n, avg, mn in rows:
        print("%-60s n=%-5d avg %9.1f us  min %9.1f us" % (name.split("(")[0].replace("void ", "")[:60], n, avg / 1e3, mn / 1e3))
