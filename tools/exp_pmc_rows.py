import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%fill_strips_v2<true>%' order by dispatch_id, counter_name"))
cur = None
for d, name, v, dur in rows:
    if d != cur:
        print("\ndispatch", d, "dur_us %.1f" % (dur / 1e3), end=": ")
        cur = d
    print("%s=%.0f" % (name.replace("SQ_", ""), v), end=" ")
print()
