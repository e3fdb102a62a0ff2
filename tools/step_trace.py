"""Per-launch durations of ONE PGD step from a rocprofv3 kernel trace (the last step of the run), in launch order:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o runc -- python3 bench.py --dtype fp32 --steps 2 --warmup 1 --eager --no_cpu_baseline --no_fft_bench --no_prof
    python tools/step_trace.py OUT > step_trace.txt

Columns: start (us, relative to the step's first kernel), duration (us), grid, kernel name."""
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "k_conv0_gn<1>" in r["Kernel_Name"] or "k_conv0_gn<2>" in r["Kernel_Name"]]
    start = first[-1] if first else 0
    t0 = int(rows[start]["Start_Timestamp"])
    for r in rows[start:]:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {dur:9.1f} {grid:>8} {r['Kernel_Name'][:110]}")


if __name__ == "__main__":
    main(sys.argv[1])
