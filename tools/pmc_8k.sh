set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc8k_$c; mkdir -p $R/gpurun_out/pmc8k_$c
  SHAPE=4320,7680,3 REPS=3 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc8k_$c -o p -- python3 $R/tools/prof_passes.py > $R/gpurun_out/pmc8k_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc8k_{c}/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"]==c and "dspfft::" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:90]].append(float(row["Counter_Value"]))
    for k,v in acc.items(): print(c, k, round(sum(v)/len(v)/1024,1), "MB per dispatch (raw KB counter)")
PY
