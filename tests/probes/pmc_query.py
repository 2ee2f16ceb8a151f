"""print avg counter values for kernels matching argv[2] from a rocprofv3 rocpd database"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "mlp_"
q = ("select substr(kernel_name,1,30), counter_name, count(*), avg(value), avg(duration) from counters_collection "
     "where kernel_name like ? group by 1,2 order by 1,2")
for r in con.execute(q, (f"%{pat}%",)):
    print(f"{r[0]:32s} {r[1]:28s} n={r[2]:3d} avg={r[3]:14.0f} dur_us={r[4]/1e3:8.1f}")
