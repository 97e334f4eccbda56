"""Gaps between the launches of the captured training step (rocprofv3 --kernel-trace database of bench.py --timed-only):
per step (a step starts with k_copy_many) the wall span, the union of the kernel intervals, and where the idle time sits
(by the kernel that precedes the gap).  usage: python scripts/prof_gaps.py <results.db>"""
import collections
import sqlite3
import sys

db = sys.argv[1]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# the step's own stream only: the upload stream's launches (id prep, the blits of the H2D copies) overlap the step and would
# count as "busy" while the step's stream idles
SIDE = ("k_batch_prep", "void k_batch_prep", "__amd_rocclr")
rows = [r for r in rows if not r[0].startswith(SIDE)]
# steps: from one k_copy_many to the next (rotating batches; a static-batch replay has none: the cross-entropy launch, one
# per step, delimits instead); keep the last 30 complete ones (graph replays)
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_copy_many")]
DELIM = "k_copy_many"
if len(starts) < 3:
    starts = [i for i, r in enumerate(rows) if "k_ce_fused" in r[0]]
    DELIM = "k_ce_fused (static replay: steps cut at the cross-entropy launch)"
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-30:]
span = busy = 0.0
by_prev = collections.Counter()
cnt_prev = collections.Counter()
overlap = 0.0
for a, b in steps:
    ks = rows[a:b]
    span += rows[b][1] - ks[0][1]
    t_end = ks[0][1]
    for i, (n, s, e) in enumerate(ks):
        if s > t_end:
            gap = s - t_end
            prev = ks[i - 1][0] if i else "(start)"
            by_prev[prev.split("(")[0][:60]] += gap
            cnt_prev[prev.split("(")[0][:60]] += 1
        else:
            overlap += min(e, t_end) - s
        busy += max(0, e - max(s, t_end))
        t_end = max(t_end, e)
    # the tail: last kernel's end to the next step's first kernel
    if rows[b][1] > t_end:
        by_prev["(step tail -> next %s)" % DELIM] += rows[b][1] - t_end
        cnt_prev["(step tail -> next %s)" % DELIM] += 1
n = float(len(steps))
print("%d steps: span %.1f us per step, kernels busy (union) %.1f us, idle %.1f us, overlapped kernel time %.1f us"
      % (len(steps), span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3, overlap / n / 1e3))
print("idle time by the kernel in front of the gap (us per step, gaps per step, us per gap):")
for k, v in by_prev.most_common(25):
    print("  %8.1f  %5.1f  %6.2f  %s" % (v / n / 1e3, cnt_prev[k] / n, v / cnt_prev[k] / 1e3, k))
