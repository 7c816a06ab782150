#!/bin/bash
# Collect the judged artefacts of one round on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh <tag>        e.g. r02d
# -> gpurun_out/<tag>_bench_n1.json          the default `python bench.py` line (HIP-event per-kernel table, live PMC traffic, extra_configs, cpu_baseline)
#    gpurun_out/<tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary of `bench.py --steps 5 --warmup 2` (same workload)
#    gpurun_out/<tag>_pmc_mfma_busy.json     SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) per kernel (own pass, --kernel-trace only)
# (copy them into profiles/ afterwards).
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2> $O/${TAG}_bench.err | grep '^{"metric' | tail -1 > $O/${TAG}_bench_n1.json
# the profiling passes run the same certified configuration with the eps (both block-tail levels' and the level the run settled on) the full run calibrated (--certify-eps: no calibration pass, no self-check launches among the steps)
EPS=$(python -c "import json;print(json.load(open('$O/${TAG}_bench_n1.json'))['precision_guarantee']['certificate']['profile_eps_arg'])")
rm -rf $O/${TAG}_prof && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras --no-stream --substreams 1 --certify-eps $EPS > /dev/null 2>&1
cp $(ls $O/${TAG}_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_prof
rm -rf $O/${TAG}_pmc_MFMA && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_MFMA -- python $R/bench.py --certify-eps $EPS --no-stream --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --substreams 1 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/${TAG}_pmc_MFMA > $O/${TAG}_pmc_mfma_raw.json
rm -rf $O/${TAG}_pmc_MFMA
python - <<PY
import json
bench = json.load(open("$O/${TAG}_bench_n1.json"))
mraw = json.load(open("$O/${TAG}_pmc_mfma_raw.json"))
busy = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py --steps 2 --warmup 1 (precision %s)" % bench["config"]["precision"],
        "definition": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE): busy cycles summed over 1024 SIMDs, active cycles summed over 8 XCDs", "kernels": {}}
for k, v in sorted(mraw.items()):
    b = v.get("SQ_VALU_MFMA_BUSY_CYCLES"); g = v.get("GRBM_GUI_ACTIVE")
    if b and g and g["sum"] > 0 and b["sum"] > 0:
        busy["kernels"][k] = {"dispatches": b["dispatches"], "mfma_busy": round(b["sum"] / (128.0 * g["sum"]), 4)}
json.dump(busy, open("$O/${TAG}_pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps(busy["kernels"]))
print(bench["value"], bench["ms_per_step"], bench["roofline"]["kernel"], bench["roofline"]["achieved"], bench["roofline"]["frac"], bench["traffic"]["measured_hbm_mb_per_step"], bench.get("cpu_baseline", {}).get("value"))
for e in bench["extra_configs"]: print(e["config"][:70], e["value"], e["ms_per_step"])
PY
rm -f $O/${TAG}_pmc_mfma_raw.json
head -14 $O/${TAG}_kernel_stats.csv | cut -c1-150
