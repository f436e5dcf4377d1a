#!/usr/bin/env python3
"""Condense rocprofv3 result databases (gpurun_out/prof/<tag>/{kt,fetch,write}/*_results.db) into a markdown
summary for profiles/.   usage: tools/rocprof_summary.py <dir-with-kt-fetch-write> <kernel-substring> <out.md> [title]"""
import glob
import sqlite3
import sys

import numpy as np


def db_of(d):
    f = glob.glob(d + "/*_results.db")
    return sqlite3.connect(f[0]) if f else None


def main():
    root, kern, outp = sys.argv[1], sys.argv[2], sys.argv[3]
    title = sys.argv[4] if len(sys.argv) > 4 else ""
    out = ["# %s\n\n" % title]
    db = db_of(root + "/kt")
    if db:
        cur = db.cursor()
        out.append("## rocprofv3 --kernel-trace --stats (top kernels)\n\n| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
        for r in cur.execute("select * from top_kernels limit 6"):
            out.append("| %s | %d | %.1f | %.3f | %.2f |\n" % (r[0][:120].replace("|", "/"), r[1], r[2], r[3], r[4]))
        rows = cur.execute("select duration,start from kernels where name like ? order by start", ("%" + kern + "%",)).fetchall()
        d = np.array([r[0] for r in rows]) / 1e3
        out.append("\nkernel `%s`: n=%d mean %.3f us, median %.3f, min %.3f, p10 %.3f, p90 %.3f\n" % (
            kern, len(d), d.mean(), np.median(d), d.min(), np.percentile(d, 10), np.percentile(d, 90)))
        st = np.array([r[1] for r in rows]); du = np.array([r[0] for r in rows])
        gaps = (st[1:] - (st[:-1] + du[:-1])) / 1e3
        gaps = gaps[(gaps > -1) & (gaps < 5)]
        if len(gaps):
            out.append("gap end(k) -> start(k+1) between consecutive launches: median %.3f us (0 means the profiler's "
                       "timestamps abut: the reported duration then includes the dispatch gap)\n" % np.median(gaps))
        r = cur.execute("select vgpr_count, accum_vgpr_count, sgpr_count, grid_x, workgroup_x, lds_size, scratch_size from kernels "
                        "where name like ? limit 1", ("%" + kern + "%",)).fetchone()
        out.append("resources: vgpr=%s agpr=%s sgpr=%s grid=%s wg=%s lds=%s scratch=%s\n" % r)
    import os
    passes = [("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")]
    passes += [(d, d[4:]) for d in sorted(os.listdir(root)) if d.startswith("pmc_")]
    for name, ctr in passes:
        db = db_of(root + "/" + name)
        if not db:
            continue
        v = np.array([x[0] for x in db.cursor().execute(
            "select value from counters_collection where kernel_name like ? and counter_name=?", ("%" + kern + "%", ctr))])
        if len(v):
            unit = "KiB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else "events"
            out.append("\n## rocprofv3 --pmc %s (separate pass)\n%s per launch of `%s`: mean %.1f %s, median %.1f, max %.1f (n=%d)\n" % (
                ctr, ctr, kern, v.mean(), unit, np.median(v), v.max(), len(v)))
    open(outp, "w").writelines(out)
    print("".join(out))


if __name__ == "__main__":
    main()
