"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per kernel launch, next to the
ALGORITHMIC bytes per launch that the same bench run computed from the GEMM descriptors.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o runc -- python3 bench.py --dtype fp32 --steps 2 --warmup 1 --no_cpu_baseline --no_fft_bench --no_prof
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o runc -- python3 bench.py --dtype fp32 --steps 2 --warmup 1 --no_cpu_baseline --no_fft_bench --no_prof
    python bench.py --dtype fp32 --steps 5 --no_cpu_baseline --no_fft_bench > bench_fp32.json          # carries all_gemm_variants[*].algorithmic_MB_per_launch
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 fp32 bench_fp32.json > profiles/<round>_hbm_traffic_pmc_fp32.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KB (x1024), and
FETCH_SIZE under-reports by 2x on gfx950 (doubled here).  The third argument is the number of steps the run executed
(warm-up + timed + the one cross-mode comparison step bench.py appends), used for the per-step columns.
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


def variant_of(kernel: str):
    """PMC kernel name -> the label bench.py's roofline uses for the same launches (None: not a GEMM)."""
    m = re.search(r"k_gemm_ring2?<(\d+), (\d+), (\d+), (\d+)", kernel)
    if m:
        return f"k_gemm_ring<{m.group(1)},{m.group(2)},{'split' if m.group(4) == '1' else 'bf16'}>"
    m = re.search(r"k_gemm_bf<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)(?:, \w+)*>", kernel)
    if m:
        bm, bn, prec = m.group(1), m.group(2), "split" if m.group(3) == "1" else "bf16"
        if bm in ("256", "192") and m.group(5) == "true" and m.group(6) == "false":
            return f"k_gemm_bf<{bm},{bn},{prec}>"
        return f"k_gemm_bf<128,{bn},{prec}>"
    m = re.search(r"k_gemm_win<(\d+)", kernel)
    if m:
        return "k_gemm_win<split>" if m.group(1) == "1" else "k_gemm_win<bf16>"
    return None


def main():
    fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    dtype = sys.argv[4] if len(sys.argv) > 4 else None
    alg = {}
    if len(sys.argv) > 5:
        with open(sys.argv[5]) as f:
            line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        roof = b["roofline"] if (dtype in (None, "fp32") or "bf16" not in b) else b["bf16"]["roofline"]
        for v in (roof or {}).get("all_gemm_variants", []):
            alg[v["kernel"]] = v.get("algorithmic_MB_per_launch")
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    kernels = []
    for name in ft:
        fb = ft[name] * 1024 * 2            # KB -> B, gfx950 FETCH_SIZE correction
        wb = wt.get(name, 0.0) * 1024
        n = fc[name]
        k = {"kernel": name, "launches_per_step": round(n / steps, 2), "fetch_GB_per_step": round(fb / steps / 1e9, 3),
             "write_GB_per_step": round(wb / steps / 1e9, 3), "hbm_bytes_per_launch": int((fb + wb) / n)}
        v = variant_of(name)
        if v:
            k["variant"] = v
            if alg.get(v):
                k["algorithmic_bytes_per_launch"] = int(alg[v] * 1e6)
                k["pmc_over_algorithmic"] = round(k["hbm_bytes_per_launch"] / (alg[v] * 1e6), 3)
        kernels.append(k)
    # launches that do not recur every step (model construction: workspace fills, weight preparation) are listed apart and kept
    # out of the per-step totals
    setup = [k for k in kernels if k["launches_per_step"] < 1.0]
    kernels = [k for k in kernels if k["launches_per_step"] >= 1.0]
    for k in setup:
        k["note"] = "setup (not per step): totals over the run, GB"
        k["fetch_GB_total"] = round(k.pop("fetch_GB_per_step") * steps, 3)
        k["write_GB_total"] = round(k.pop("write_GB_per_step") * steps, 3)
    kernels.sort(key=lambda k: -(k["fetch_GB_per_step"] + k["write_GB_per_step"]))
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --dtype <dtype> --steps 2 "
                   "--warmup 1 --no_cpu_baseline --no_fft_bench --no_prof`; KB units x1024; FETCH_SIZE doubled per the gfx950 correction "
                   "(MI355X_MICROARCH.md, HBM).  algorithmic_bytes_per_launch: every operand / result byte of the GEMM descriptors once "
                   "(csrc/gemm.hip, paa_prof), averaged over the variant's launches in a bench run of the same commit; several kernel "
                   "instantiations can share one variant label, their launches then average over different shapes.",
           "dtype": dtype,
           "total_fetch_GB_per_step": round(sum(k["fetch_GB_per_step"] for k in kernels), 2),
           "total_write_GB_per_step": round(sum(k["write_GB_per_step"] for k in kernels), 2),
           "kernels": kernels, "setup_kernels": setup}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
