"""rocprofv3 --kernel-trace database -> average duration of a kernel per grid size.  usage: kernel_by_grid.py run.db KERNEL_LIKE"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
cur.execute("select grid_x, count(*), avg(duration), min(duration) from kernels where name like ? group by grid_x order by grid_x", ("%" + sys.argv[2] + "%",))
for g, n, a, m in cur.fetchall():
    print("grid_x %8d  calls %4d  avg %9.2f us  min %9.2f us" % (g, n, a / 1e3, m / 1e3))
