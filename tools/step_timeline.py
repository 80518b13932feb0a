"""Print one training step of a rocprofv3 --kernel-trace CSV as a timeline (start, duration, queue, kernel).
usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
busy = 0.0
print(len(step), "kernels; span us", (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3)
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    print("%8.1f %7.1f q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r.get("Queue_Id", "?"), n))
print("sum of durations", round(busy, 1))
