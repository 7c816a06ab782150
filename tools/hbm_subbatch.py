"""Developer tool: HBM bytes per step (rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and step time of the
batch-32 headline configuration when the layer loop runs over sub-batches:
    full       one group of 32 pairs                       (--substreams 1)
    conc2      two groups of 16 pairs on two streams       (--substreams 2, the bench default)
    serial2    two groups of 16 pairs, one after the other (--substreams 2 --debug-variant 26:1): ~125 MB of live activations per group
    serial4    four groups of 8 pairs, one after the other (~60 MB live; launches of 8 pairs leave CUs idle)
usage: python tools/hbm_subbatch.py [configs...]   (run through gpurun; writes gpurun_out/hbm_subbatch.json)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
CONFIGS = {"full": ["--substreams", "1"], "conc2": ["--substreams", "2"], "serial2": ["--substreams", "2", "--debug-variant", "26:1"],
           "serial4": ["--substreams", "4", "--debug-variant", "26:1"], "conc4": ["--substreams", "4"]}
COMMON = ["--no-cpu-baseline", "--no-traffic", "--no-extras", "--no-stream", "--no-rccl-check"]


def pmc(flags, counter, steps=2, warmup=1):
    work = tempfile.mkdtemp(prefix="gn_hbm_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", work, "--", sys.executable, BENCH,
               "--steps", str(steps), "--warmup", str(warmup)] + COMMON + flags
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
        per = {}
        for f in glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name") or ""
                    if "gn::" not in k:
                        continue
                    k = k.replace("void ", "").replace("gn::(anonymous namespace)::", "").replace("gn::", "").split("(")[0]
                    e = per.setdefault(k, [0.0, 0])
                    e[0] += float(row.get("Counter_Value") or 0)
                    e[1] += 1
        return per, steps + warmup, r.returncode
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    names = sys.argv[1:] or ["full", "conc2", "serial2", "serial4"]
    out = {}
    for name in names:
        flags = CONFIGS[name]
        r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5"] + COMMON + flags, cwd="/tmp", capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
        ms = json.loads(line[-1])["ms_per_step"] if line else None
        fetch, n, rc1 = pmc(flags, "FETCH_SIZE")
        write, _, rc2 = pmc(flags, "WRITE_SIZE")
        kern = {}
        for k in set(fetch) | set(write):
            f = fetch.get(k, [0.0, 0]); w = write.get(k, [0.0, 0])
            kern[k] = {"launches_per_step": round(max(f[1], w[1]) / n, 2), "hbm_mb_per_step": round((2.0 * f[0] + w[0]) * 1024.0 / n / 1e6, 1)}
        total = sum(v["hbm_mb_per_step"] for v in kern.values())
        top = dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_mb_per_step"])[:6])
        out[name] = {"flags": flags, "ms_per_step": ms, "hbm_mb_per_step": round(total, 1), "rc": [r.returncode, rc1, rc2], "top_kernels": top}
        print(name, json.dumps(out[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "hbm_subbatch.json"), "w") as f:
        json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 3 steps of bench.py; bytes = 2 x FETCH_SIZE + WRITE_SIZE; "
                             "ms_per_step from an unprofiled 20-step run of the same flags", "configs": out}, f, indent=1)


if __name__ == "__main__":
    main()
