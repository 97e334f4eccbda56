"""Summarise a rocprofv3 --kernel-trace rocpd database: per-kernel totals (like --stats)."""
import sqlite3, sys
db = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/r1_results.db"
con = sqlite3.connect(db)
cur = con.cursor()
# steps in the trace: every step (warm-up, timed, instrumented passes of bench.py) launches the Adam kernel once
steps = float(sys.argv[2]) if len(sys.argv) > 2 else float(
    cur.execute("select count(*) from kernels where name like '%k_adam%'").fetchone()[0] or 1)
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels where name not like 'k_spin%' group by name order by 3 desc").fetchall()   # k_spin: bench.py's measurement aid
tot = sum(r[2] for r in rows)
print("kernel time total %.3f ms over %d launches (%.3f ms / step over %g steps)" %
      (tot / 1e6, sum(r[1] for r in rows), tot / 1e6 / steps, steps))
print("%7s %10s %7s %10s %9s %9s  %s" % ("%", "total_ms", "calls", "avg_us", "min_us", "max_us", "kernel"))
for r in rows:
    print("%6.2f%% %10.3f %7d %10.1f %9.1f %9.1f  %s" %
          (100 * r[2] / tot, r[2] / 1e6, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[0][:110]))
