"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last few optimizer launches, what ran concurrently with them:
usage: timeline_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
adam = [k for k in ks if "adam" in k[2]]
print("kernels", len(ks), "adam launches", len(adam))
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n[:60]
for a in adam[-4:]:
    print(f"ADAM {short(a[2])} q{a[3]}  dur {(a[1]-a[0])/1e3:8.1f} us")
    ov = [k for k in ks if k is not a and k[0] < a[1] and k[1] > a[0]]
    for k in ov[:40]:
        print(f"    {(k[0]-a[0])/1e3:9.1f} .. {(k[1]-a[0])/1e3:9.1f} us  dur {(k[1]-k[0])/1e3:7.1f}  q{k[3]}  {short(k[2])}")
# typical durations by kernel name over the whole trace
by = {}
for k in ks:
    by.setdefault(short(k[2]), []).append((k[1] - k[0]) / 1e3)
print("median durations:")
for n, v in sorted(by.items(), key=lambda x: -sum(x[1]))[:14]:
    v.sort(); print(f"  {n:60s} n={len(v):5d} med {v[len(v)//2]:8.1f} us")
