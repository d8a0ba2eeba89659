"""Summarises rocprofv3 rocpd sqlite outputs (gpurun_out/*/…_results.db) into small text files for profiles/.
  python tools/rocpd_summary.py stats  <db> <out.csv>     -- --kernel-trace --stats : per-kernel calls / total / average (us)
  python tools/rocpd_summary.py pmc    <db> <out.csv>     -- --pmc X : per-kernel average counter value per launch
"""
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    return '"' + n + '"' if "," in n else n      # template argument lists contain commas


def main():
    mode, db, out = sys.argv[1:4]
    c = sqlite3.connect(db)
    with open(out, "w") as f:
        if mode == "stats":
            f.write("kernel,calls,total_us,avg_us,percent\n")
            for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                f.write(f"{short(name)},{calls},{total:.3f},{avg:.3f},{pct:.2f}\n")
        else:
            f.write("kernel,counter,launches,avg_value_per_launch,avg_duration_us\n")
            q = ("select kernel_name,counter_name,count(*),avg(value),avg(duration)/1000.0 from counters_collection "
                 "group by kernel_name,counter_name order by 1,2")
            for name, ctr, n, v, d in c.execute(q):
                f.write(f"{short(name)},{ctr},{n},{v:.3f},{d:.3f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
