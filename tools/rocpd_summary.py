"""Turn rocprofv3's rocpd database (its default output here) into small CSV summaries for profiles/.
usage: rocpd_summary.py <results.db> <out.csv>   -- kernel time per (kernel, grid), or counter values when the
database holds a --pmc pass."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = []
n_pmc = c.execute("select count(*) from counters_collection").fetchone()[0]
if n_pmc:
    q = ("select kernel_name, grid_size, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
         "from counters_collection group by kernel_name, grid_size, counter_name order by avg(duration)*count(*) desc")
    hdr = ["kernel", "grid_size", "counter", "dispatches", "avg_value", "min_value", "max_value", "avg_duration_ns"]
else:
    q = ("select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration) "
         "from kernels group by name, grid_x order by sum(duration) desc")
    hdr = ["kernel", "grid_size", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "median_ns", "calls_over_1s"]
try:
    rows = list(c.execute(q))
except sqlite3.OperationalError as e:
    cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
    raise SystemExit("unexpected schema: %s; kernels columns: %s" % (e, cols))
if not n_pmc:   # median and the number of outliers (a launch stalled by the profiler counts seconds)
    import statistics
    full = []
    for r in rows:
        d = [x[0] for x in c.execute("select duration from kernels where name = ? and grid_x = ?", (r[0], r[1]))]
        full.append(tuple(r) + (statistics.median(d), sum(1 for x in d if x > 1e9)))
    rows = full
with open(out, "w", newline="") as f:
    w = csv.writer(f); w.writerow(hdr); w.writerows(rows)
for r in rows[:12]:
    print(r)
