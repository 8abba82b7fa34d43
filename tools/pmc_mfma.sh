# MFMA utilisation of the GEMM-shaped paths (zoom's basis product, applybasis' partial sums, a plain GEMM): rocprofv3 --pmc over
# tools/prof_mfma.py (or the script given as $1, e.g. tools/bench_applybasis_blocks.py), summarised per kernel and grid size into gpurun_out/mfma_pmc.json
# (gpurun_out/mfma_pmc_<script>.json for $1).  Run from the repo root on the GPU box.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
SCRIPT=${1:-tools/prof_mfma.py}
export PMC_OUT=$R/gpurun_out/mfma_pmc$( [ -n "$1" ] && echo _$(basename $1 .py) ).json
rm -rf $R/gpurun_out/pmc_mfma; mkdir -p $R/gpurun_out/pmc_mfma
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -o p -- python3 $R/$SCRIPT > $R/gpurun_out/pmc_mfma.log 2>&1
python3 - <<'PY'
import csv, glob, os, json, collections
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
f = glob.glob(f"{R}/gpurun_out/pmc_mfma/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "gemm" not in k and "zoom" not in k and "applybasis" not in k and "ab_" not in k:
        continue
    import re
    key = f'{k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]} grid={row["Grid_Size"]}'
    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python3 tools/prof_mfma.py (tools/pmc_mfma.sh)",
       "note": "per-dispatch averages; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs), the gfx94x MfmaUtil formula rocprofv3 falls back to on gfx950",
       "kernels": {}}
for key, c in acc.items():
    d = {n: sum(v) / len(v) for n, v in c.items()}
    d["calls"] = len(c["GRBM_GUI_ACTIVE"])
    if d.get("GRBM_GUI_ACTIVE"):
        d["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    out["kernels"][key] = d
json.dump(out, open(os.environ["PMC_OUT"], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, "calls", d["calls"], "mfma_util", d.get("mfma_util"))
PY
