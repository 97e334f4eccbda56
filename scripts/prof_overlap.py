"""Does a kernel overlap the rest of the step?  For every launch of KERNEL in a rocprofv3 --kernel-trace database:
its start relative to the previous kernel's end, its duration and how much of it ran beside other kernels."""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
for i, (n, s, e) in enumerate(rows):
    if pat not in n:
        continue
    others = [(n2, s2, e2) for (n2, s2, e2) in rows[max(0, i - 6):i + 8] if (n2, s2, e2) != (n, s, e)]
    ov = sum(max(0, min(e, e2) - max(s, s2)) for _, s2, e2 in others)
    prev_end = max([e2 for _, s2, e2 in others if s2 <= s] or [s])
    nxt = min([s2 for _, s2, e2 in others if s2 >= s] or [e])
    print("%-28s dur %6.1f us  overlapped %6.1f us  starts %7.1f us after prev end  next kernel starts %7.1f us after its start (%s)"
          % (n[:28], (e - s) / 1e3, ov / 1e3, (s - prev_end) / 1e3, (nxt - s) / 1e3,
             [x[0][:20] for x in others if x[1] >= s][:1]))

# sequence view: what happens between the end of one step (its k_adam) and the first GEMM of the next
if len(sys.argv) > 3 and sys.argv[3] == "seq":
    adam = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
    for a in adam[8:11] + adam[-12:-10]:
        t0 = rows[a][2]
        print("--- after k_adam (ended at 0):")
        for n, s, e in rows[a + 1:a + 9]:
            print("   %+8.1f us .. %+8.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:60]))
