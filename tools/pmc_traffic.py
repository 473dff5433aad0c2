#!/usr/bin/env python3
"""gpurun_out/profiles_raw (tools/make_profiles.sh) -> profiles/<tag>_*: kernel stats csv + summary, and the per-kernel
HBM traffic json that bench.py reports as roofline.traffic.  Usage: tools/pmc_traffic.py <tag>"""
import collections
import csv
import glob
import json
import shutil
import subprocess
import sys

tag = sys.argv[1]
raw = "gpurun_out/profiles_raw"
stats = glob.glob(raw + "/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"profiles/{tag}_bench_steps10_kernel_stats.csv")
with open(f"profiles/{tag}_bench_steps10_summary.txt", "w") as f:
    f.write(subprocess.run([sys.executable, "tools/prof_summary.py", raw + "/trace", "30"], capture_output=True, text=True).stdout)
    f.write("\n" + [l for l in open(raw + "/trace.log") if l.startswith("{")][-1])

STEPS = 3 + 1 + 3          # warmup + timed + per-stage timing pass of bench.py --steps 3 --warmup 1


def per_kernel(d, counter):
    f = glob.glob(raw + f"/{d}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return tot, n


def traffic(fetch_dir, write_dir, workload, path):
    rd, nrd = per_kernel(fetch_dir, "FETCH_SIZE")
    wr, _ = per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        kernels[k] = {"launches_per_step": len(nrd.get(k, ())) / STEPS,
                      "read_bytes_per_step": 2 * rd.get(k, 0.0) * 1024 / STEPS,
                      "write_bytes_per_step": wr.get(k, 0.0) * 1024 / STEPS}
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python bench.py --steps 3 --warmup 1 "
                   f"--cpu-baseline skip` ({workload}, N=16777216, G=512). Counter unit KiB. read_bytes = 2 * FETCH_SIZE * 1024 "
                   "(gfx950 counts a wide coalesced read at half), write_bytes = WRITE_SIZE * 1024. per_step = sum over all "
                   f"launches / {STEPS} steps.", "steps": STEPS, "kernels": kernels}
    json.dump(out, open(path, "w"), indent=1)
    tot = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for k, v in kernels.items() if "pack" not in k and "cell_keys" not in k)
    print(f"{workload}: HBM traffic per step: {tot / 1e9:.2f} GB")
    for k, v in sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes_per_step"] + kv[1]["write_bytes_per_step"]))[:16]:
        print(f"{k[:70]:70s} rd {v['read_bytes_per_step'] / 1e6:9.1f} MB  wr {v['write_bytes_per_step'] / 1e6:9.1f} MB  x{v['launches_per_step']:.2f}")


traffic("fetch", "write", "two_galaxy", f"profiles/{tag}_pmc_traffic.json")
if glob.glob(raw + "/fetch_uniform/**/*counter_collection.csv", recursive=True):
    traffic("fetch_uniform", "write_uniform", "uniform", f"profiles/{tag}_pmc_traffic_uniform.json")
