cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_packed_ef.py -m gpu -q 2>&1 | tail -3
bash tools/prof_ef_s2.sh r04h 2>&1 | grep "ms/step\|^=="
python - <<'PY'
import csv,re
for f in ("ef_s2","ef_uniform_16m","ef_s1"):
    for r in list(csv.reader(open(f"gpurun_out/r04h/{f}_kernel_stats.csv")))[1:]:
        m=re.search(r"(k_ef\w+(<[^>]*>)?)", r[0])
        if m: print(f, m.group(1).ljust(36), r[1], r[2], r[3])
PY
