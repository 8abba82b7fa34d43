# HBM traffic per launch of the four passes of the 4K roundtrip: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# separate passes over tools/prof_passes.py (MI355X_MICROARCH.md, HBM section), summarised into profiles-style JSON.
# Run on the GPU box from the repo root: bash tools/pmc_traffic.sh   (writes gpurun_out/traffic.json)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc4k_$c; mkdir -p $R/gpurun_out/pmc4k_$c
  REPS=6 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc4k_$c -o p -- python3 $R/tools/prof_passes.py > $R/gpurun_out/pmc4k_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
# kernel names end in <Spec, KIND, PLAIN>(...) since round 3
names = {("row_spec_kernel", ", 0, true>("): "row_redft10", ("col_spec_kernel", ", 0, true>("): "col_redft10", ("row_spec_kernel", ", 1, true>("): "row_redft01", ("col_spec_kernel", ", 1, true>("): "col_redft01"}
raw = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmc4k_{c}/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != c: continue
        for (kn, tag), key in names.items():
            if kn in row["Kernel_Name"] and tag in row["Kernel_Name"]:
                acc[key].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        raw[k][c + "_KB"] = round(sum(v) / len(v), 1)
# gfx950: FETCH_SIZE reports half of the bytes of wide (12-16 B/lane) streaming reads -> x2; WRITE_SIZE exact
tot = {k: (2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024 for k, v in raw.items()}
dom = max(tot, key=tot.get)
import sys
sys.path.insert(0, R)
plan_lines = None
try:
    import subprocess
    plan_lines = subprocess.check_output([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from dspfun_amd import Plan, REDFT10; print(Plan.image(2160, 3840, 3, REDFT10).describe())" % R]).decode().splitlines()
    plan_lines = [ln.strip() for ln in plan_lines if ln.startswith("axis")]
except Exception as e:
    print("could not record the plan description:", e)
out = {"round": os.environ.get("ROUND", "r06"), "plan": plan_lines, "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/prof_passes.py, one 3840x2160x3 frame (tools/pmc_traffic.sh)",
       "raw_KB_per_dispatch": raw,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (12-16 B/lane) streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": {k: int(v) for k, v in tot.items()},
       "dominant_kernel": "col_redft01", "dominant_kernel_hbm_bytes_per_launch": int(tot["col_redft01"])}
json.dump(out, open(f"{R}/gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out["hbm_bytes_per_launch"]))
PY
