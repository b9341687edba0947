"""Summarise a rocprofv3 --pmc results .db: per dispatch of kernels matching a substring, summed counter values."""
import glob, sqlite3, sys, collections
d, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fgvc")
for f in glob.glob(d + "/**/*results.db", recursive=True):
    c = sqlite3.connect(f).cursor()
    rows = c.execute("select dispatch_id, kernel_name, counter_name, sum(value), min(start), max(end) from counters_collection "
                     "where kernel_name like ? group by dispatch_id, counter_name order by dispatch_id", (f"%{filt}%",)).fetchall()
    per = collections.OrderedDict()
    for disp, k, cn, v, s, e in rows:
        per.setdefault((disp, k[:50], (e - s) / 1e3), {})[cn] = v
    names = sorted({cn for v in per.values() for cn in v})
    print("dispatch us " + " ".join(names))
    for (disp, k, us), v in per.items():
        print(f"{disp:5d} {us:9.1f} " + " ".join(f"{v.get(n, 0):.4g}" for n in names))
