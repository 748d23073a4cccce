cd $GRAFT_REPO_ROOT
for i in 1 2; do
(time python bench.py --no-s2 > gpurun_out/b$i.json 2> gpurun_out/b$i.err) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open("gpurun_out/b$i.json").read().strip().split("\n")[-1])
print("S1", round(d["value"]/1e6,2), round(d["ms_per_step"],3))
for k,v in d["extra"].items():
    if isinstance(v,dict) and "ms_per_step" in v:
        print(k, round(v["ms_per_step"],3), "mean", round(v["ms_per_step_mean"],3), {a:round(b,3) for a,b in v["kernel_ms"].items()}, "mean", {a:round(b,3) for a,b in v["kernel_ms_mean"].items()}, "G ids/s", round(v["ids_per_s"]/1e9,3), "frac", round(v["frac_of_hbm_peak"],4))
PY
done
