"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) result database: per-kernel launch statistics and, when
present, per-kernel PMC counter sums.  Usage: python tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def summarise(path):
    con = sqlite3.connect(path)
    out = ["# %s" % path]
    rows = con.execute(
        """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size),
                  max(d.workgroup_size_x), max(d.grid_size_x)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out.append("KERNEL_DISPATCH stats (times in ns)")
    out.append("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,arch_vgpr,accum_vgpr,sgpr,LDS_bytes,scratch_bytes,wg_size,grid_size")
    for r in rows:
        out.append("%s,%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s,%s,%s" % (r[0].replace(".kd", ""), r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                                   r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    pm = con.execute(
        """select s.kernel_name, p.name, count(*), sum(e.value), avg(e.value)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, p.name order by 4 desc""").fetchall()
    if pm:
        out.append("PMC counters per kernel (sum / average per dispatch)")
        out.append("Name,Counter,Dispatches,Sum,AveragePerDispatch")
        for r in pm:
            out.append("%s,%s,%d,%.6g,%.6g" % (r[0].replace(".kd", ""), r[1], r[2], r[3], r[4]))
    return "\n".join(out)


def timeline(path):
    """--timeline <results.db>: one line per kernel dispatch (start and end in ms since the first dispatch, queue, grid) in start order --
    which launches overlap, and where a stream waits."""
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = con.execute("select s.kernel_name, d.start, d.end, d.%s, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start" % q).fetchall()
    t0 = rows[0][1] if rows else 0
    out = ["name,start_ms,end_ms,duration_ms,queue,grid"]
    for r in rows:
        nm = r[0].replace(".kd", "")
        nm = nm[:60]
        out.append("%s,%.3f,%.3f,%.3f,%s,%s" % (nm, (r[1] - t0) * 1e-6, (r[2] - t0) * 1e-6, (r[2] - r[1]) * 1e-6, r[3], r[4]))
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
        print(timeline(sys.argv[2]))
        sys.exit(0)
    for p in sys.argv[1:]:
        print(summarise(p))
        print()
