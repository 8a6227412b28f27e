"""Print the kernel + memory-copy timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd .db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
ev = []
for name, q, s, e, g in cur.execute("select name,queue_id,start,end,grid_x from kernels"):
    ev.append((s, e, 'K', name.split('(')[0][-32:], q, g))
for name, q, s, e, sz in cur.execute("select name,queue_id,start,end,size from memory_copies"):
    ev.append((s, e, 'C', name.replace("MEMORY_COPY_", ""), q, sz))
ev.sort()
first = [i for i, x in enumerate(ev) if 'viterbi_kernel' in x[3]][0]
t0 = ev[first][0]
for x in ev[first + skip: first + skip + limit]:
    print(f"{(x[0]-t0)/1e6:9.3f} {(x[1]-t0)/1e6:9.3f} dur {(x[1]-x[0])/1e6:7.3f} {x[2]} q{x[4]} {x[3]} {x[5]}")
