# Where a kernel's wave cycles go: rocprofv3 --pmc over one command, the SQ wait / active counters summed per kernel.
#   tools/pmc_sq.sh <tag> <command ...>      (run from the repo root on the GPU box; writes gpurun_out/pmc_sq_<tag>.txt)
# rocprofv3 runs from /tmp: arguments that name files under the repo (tools/x.py) are made absolute first; a failed pass is reported, not hidden.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
args=()
for a in "$@"; do if [ -e "$R/$a" ] && [ "${a#/}" = "$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done
set -- "${args[@]}"
rm -rf $R/gpurun_out/pmc_sq_$tag; mkdir -p $R/gpurun_out/pmc_sq_$tag
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_sq_$tag/a -o p -- "$@" > $R/gpurun_out/pmc_sq_$tag/a.log 2>&1 || echo "pmc_sq: pass a FAILED (see gpurun_out/pmc_sq_$tag/a.log)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq_$tag/b -o p -- "$@" > $R/gpurun_out/pmc_sq_$tag/b.log 2>&1 || echo "pmc_sq: pass b FAILED (see gpurun_out/pmc_sq_$tag/b.log)"
python3 - "$R/gpurun_out/pmc_sq_$tag" > $R/gpurun_out/pmc_sq_$tag.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-90:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    d = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, "dispatches", max(len(v) for v in c.values()))
    for n in sorted(d):
        print("   %-24s %14.0f" % (n, d[n]))
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if n in d:
                print("   %-24s / SQ_WAVE_CYCLES = %.3f" % (n, d[n] / wc))
PY
cat $R/gpurun_out/pmc_sq_$tag.txt
