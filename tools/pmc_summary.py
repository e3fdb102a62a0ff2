"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per kernel launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o runc -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_prof
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o runc -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_prof
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 3 > profiles/<round>_hbm_traffic_pmc.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KB (x1024), and
FETCH_SIZE under-reports by 2x on gfx950 (doubled here).  The third argument is the number of steps the run executed
(warm-up + timed), used for the per-step columns.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\([^()]*\)$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).strip()
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    kernels = []
    for name in ft:
        fb = ft[name] * 1024 * 2            # KB -> B, gfx950 FETCH_SIZE correction
        wb = wt.get(name, 0.0) * 1024
        n = fc[name]
        kernels.append({"kernel": name, "launches_per_step": round(n / steps, 2), "fetch_GB_per_step": round(fb / steps / 1e9, 3),
                        "write_GB_per_step": round(wb / steps / 1e9, 3), "hbm_bytes_per_launch": int((fb + wb) / n)})
    kernels.sort(key=lambda k: -(k["fetch_GB_per_step"] + k["write_GB_per_step"]))
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --steps 2 "
                   "--warmup 1 --no_cpu_baseline --no_prof`; KB units x1024; FETCH_SIZE doubled per the gfx950 correction "
                   "(MI355X_MICROARCH.md, HBM)",
           "total_fetch_GB_per_step": round(sum(k["fetch_GB_per_step"] for k in kernels), 2),
           "total_write_GB_per_step": round(sum(k["write_GB_per_step"] for k in kernels), 2),
           "kernels": kernels}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
