#!/bin/bash
# Collect the judged artefacts of one round on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh <tag>        e.g. r01d
# -> gpurun_out/<tag>_bench_n1.json, <tag>_kernel_stats.csv, <tag>_pmc_hbm_traffic.json
# (copy them into profiles/ afterwards).  PMC counters are collected in their own passes with --kernel-trace only.
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench_n1.json
rm -rf $O/${TAG}_prof && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/${TAG}_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${TAG}_pmc_$c && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
# matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs)
rm -rf $O/${TAG}_pmc_MFMA && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_MFMA -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/${TAG}_pmc_MFMA > $O/${TAG}_pmc_mfma_raw.json
python $R/tools/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_raw.json
python - <<PY
import json
raw = json.load(open("$O/${TAG}_pmc_raw.json"))
bench = json.load(open("$O/${TAG}_bench_n1.json"))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline (3 steps, precision %s), MI355X" % bench["config"]["precision"],
       "units": "KB per dispatch; corrected_fetch = 2 x FETCH_SIZE for wide coalesced reads on gfx950 (MI355X_MICROARCH.md; calibrated on "
                "k_ln_gelu in round 1: 65587 KB raw for 131072 KB read)",
       "kernels": {}}
tot_b = tot_n = 0
for k, v in sorted(raw.items()):
    f = v.get("FETCH_SIZE", {"avg": 0, "dispatches": 0}); w = v.get("WRITE_SIZE", {"avg": 0, "dispatches": 0})
    n = max(f["dispatches"], w["dispatches"])
    b = int((2 * f["avg"] + w["avg"]) * 1024)
    out["kernels"][k] = {"dispatches": n, "fetch_kb_raw_avg": round(f["avg"], 1), "write_kb_avg": round(w["avg"], 1), "hbm_bytes_per_launch_corrected": b}
    if k.startswith("k_gemm"):
        tot_b += b * n; tot_n += n
out["k_gemm_f32_all_variants"] = {"dispatches": tot_n, "hbm_bytes_per_launch_corrected": int(tot_b / max(tot_n, 1)),
                                  "note": "call-weighted over all GEMM launches of a step (75 per step)"}
json.dump(out, open("$O/${TAG}_pmc_hbm_traffic.json", "w"), indent=1)
mraw = json.load(open("$O/${TAG}_pmc_mfma_raw.json"))
busy = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py --steps 2 --warmup 1 (precision %s)" % bench["config"]["precision"],
        "definition": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE): busy cycles summed over 1024 SIMDs, active cycles summed over 8 XCDs", "kernels": {}}
for k, v in sorted(mraw.items()):
    b = v.get("SQ_VALU_MFMA_BUSY_CYCLES"); g = v.get("GRBM_GUI_ACTIVE")
    if b and g and g["sum"] > 0 and b["sum"] > 0:
        busy["kernels"][k] = {"dispatches": b["dispatches"], "mfma_busy": round(b["sum"] / (128.0 * g["sum"]), 4)}
json.dump(busy, open("$O/${TAG}_pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps(busy["kernels"]))
print(json.dumps(out["k_gemm_f32_all_variants"]))
PY
cat $O/${TAG}_bench_n1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline']['value'])"
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-150
