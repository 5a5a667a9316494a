"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite database):
    python tools/rocprof_gaps.py <results.db> [min_kernel_us]
Prints the span of the trace's busiest contiguous stretch, the summed kernel time and the summed gaps, plus a histogram of gaps -
what the hipGraph's kernel-to-kernel hand-over costs per step."""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    if not rows:
        print("no kernels"); return
    # the sampling loop = the stretch with the most kernel time between pauses longer than 50 ms (model build, weight upload and
    # the output gather lie outside)
    segs, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][1] > 50e6:
            segs.append(cur); cur = []
        cur.append(r)
    segs.append(cur)
    rows = max(segs, key=lambda sg: sum(e - s for s, e, _ in sg))
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = [max(0, rows[i][0] - rows[i - 1][1]) for i in range(1, len(rows))]
    overlap = sum(max(0, rows[i - 1][1] - rows[i][0]) for i in range(1, len(rows)))
    print(f"kernels {len(rows)}  span {span / 1e6:.1f} ms  kernel time {busy / 1e6:.1f} ms  gaps {sum(gaps) / 1e6:.1f} ms "
          f"({100.0 * sum(gaps) / span:.2f} % of the span)  overlap {overlap / 1e6:.2f} ms")
    edges = [0, 1e3, 2e3, 4e3, 8e3, 16e3, 64e3, 1e6, 1e12]
    for a, b in zip(edges, edges[1:]):
        sel = [g for g in gaps if a <= g < b]
        print(f"  gap {a / 1e3:7.0f} .. {b / 1e3:9.0f} us: {len(sel):6d} gaps, {sum(sel) / 1e6:8.2f} ms")
    big = sorted(((rows[i][0] - rows[i - 1][1], rows[i - 1][2][:50], rows[i][2][:50]) for i in range(1, len(rows))), reverse=True)[:8]
    for g, a, b in big:
        print(f"  {g / 1e3:9.1f} us between {a} -> {b}")


if __name__ == "__main__":
    main(sys.argv[1])
