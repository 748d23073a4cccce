# rocprofv3 kernel stats of any command, printed as "kernel  calls  avg_ns": usage: bash tools/kt_cmd.sh <tag> <command ...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=$1; shift; mkdir -p gpurun_out/$T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/kt -o g -- "$@" > gpurun_out/$T/kt.out 2> gpurun_out/$T/kt.err
find gpurun_out/$T/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/$T/kstats.csv \;
rm -rf gpurun_out/$T/kt
python - <<PY
import csv, re
for r in list(csv.reader(open("gpurun_out/$T/kstats.csv")))[1:]:
    m = re.search(r"k_\\w+(<[^>]*>)?", r[0])
    if m and ("vidc" in r[0] or "anonymous" in r[0]): print(m.group(0).ljust(56), r[1].rjust(6), "%10.1f us" % (float(r[3]) / 1e3))
PY
