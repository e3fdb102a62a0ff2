"""Idle time between the kernels of one PGD step from a rocprofv3 kernel trace of a graph-replay run:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o runc -- python3 bench.py --dtype fp32 --steps 4 --warmup 2 --no_cpu_baseline --no_fft_bench --no_prof
    python tools/step_gaps.py OUT

For the last complete step: launches, sum of kernel durations, wall span, sum of positive gaps (start[i+1] - end[i]), the gap
histogram, and the ten largest gaps with the kernels on either side."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "k_conv0_gn<1>" in r["Kernel_Name"] or "k_conv0_gn<2>" in r["Kernel_Name"]]
    if len(first) < 3:
        print("not enough steps in the trace"); return
    a, b = first[-3], first[-2]                 # a complete step in the middle of the timed region
    step = rows[a:b]
    st = [int(r["Start_Timestamp"]) for r in step] + [int(rows[b]["Start_Timestamp"])]
    en = [int(r["End_Timestamp"]) for r in step]
    dur = sum(e - s for s, e in zip(st, en)) / 1e3
    span = (st[-1] - st[0]) / 1e3
    gaps = [(st[i + 1] - en[i]) / 1e3 for i in range(len(step))]
    pos = sum(g for g in gaps if g > 0)
    print(f"launches {len(step)}  kernel time {dur:.1f} us  span (first start -> next step's first start) {span:.1f} us  positive gaps {pos:.1f} us  overlaps {sum(g for g in gaps if g < 0):.1f} us")
    h = collections.Counter(min(int(g), 20) for g in gaps)
    print("gap histogram (us -> launches):", dict(sorted(h.items())))
    order = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:10]
    for i in order:
        nxt = step[i + 1]["Kernel_Name"] if i + 1 < len(step) else rows[b]["Kernel_Name"]
        print(f"  {gaps[i]:7.1f} us after {step[i]['Kernel_Name'][:70]}  before {nxt[:60]}")
    by = collections.defaultdict(float)
    for i, g in enumerate(gaps):
        by[step[i]["Kernel_Name"][:60]] += max(g, 0)
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  gap after {k}: {v:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
