# end-of-round evidence run: tests, smoke, default bench, secondary workloads, rocprofv3 kernel stats
set -x
cd $GRAFT_REPO_ROOT
R=${1:-r01e}
mkdir -p gpurun_out/$R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/$R/pytest_gpu.txt
VIDC_FORCE_GENERAL=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_force_general.txt
VIDC_NO_LANE=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_no_lane.txt
VIDC_FORCE_LANE=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_force_lane.txt
VIDC_OLD_U=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_old_u.txt
VIDC_FULL_PREPASS=1 VIDC_NO_LANE_REG=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_full_prepass_no_lane_reg.txt
VIDC_FORCE_GRP=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_force_grp.txt
GPU_MAX_HW_QUEUES=8 VIDC_WIDE_STREAMS=1 timeout 1200 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_wide.txt
VIDC_NO_LANE_PAIR=1 VIDC_FORCE_LANE=1 timeout 900 python -m pytest tests/test_gpu_roc.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_no_lane_pair.txt
# round 4: the loop form of the bucket-row lane decoders + 256 buckets for 1025..2048 ids; per-list classification instead of the length order
VIDC_LANE_LOOP=1 VIDC_NO_LANE128=1 VIDC_FORCE_LANE=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_lane_loop.txt
VIDC_NO_AVX2=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_packed_ef.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_no_avx2.txt
VIDC_NO_LENGTH_CLASSES=1 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_containers.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_no_length_classes.txt
# round 5: 65..256-id lists on lane pairs must not change a bit (the look-ahead / chain-priority switches of that round are gone: HISTORY.md)
VIDC_PAIR_MIN=64 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_r5_switches.txt
# round 5, last session: the forms the new defaults replaced (whole-row b2 loads, lane-decoder rows on 16-byte boundaries, no class priority) and
# the sized row load forced for every b2 launch must not change a bit either
VIDC_B2_MASK=0 VIDC_LANE_ALIGN=4 VIDC_ENC_PRIO=0 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_r5_old_defaults.txt
VIDC_B2_MASK=1 VIDC_ENC_PRIO=7 timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_r5_mask_prio.txt
VIDC_POOL_POISON=1 timeout 1200 python -m pytest tests/test_gpu_roc.py tests/test_gpu_packed_ef.py tests/test_gpu_containers.py -m gpu -q 2>&1 | tail -3 > gpurun_out/$R/pytest_pool_poison.txt
timeout 300 python tools/bench_search_paths.py 2>&1 | tail -12 > gpurun_out/$R/search_paths.txt
# S2 decoded 100 times per mode and compared with the first decode (the list-level flake hunt of round 3)
(GPU_MAX_HW_QUEUES=8 NQS=4,8,5,6 ITERS=100 timeout 200 python tools/repro_s2b.py "X=1" 2>&1 | grep "differs\|it 99" | cut -c1-160; GPU_MAX_HW_QUEUES=4 NQS=3 ITERS=100 timeout 200 python tools/repro_s2b.py "X=1" 2>&1 | grep "differs\|it 99" | cut -c1-160) > gpurun_out/$R/s2_repeated_decodes.txt
timeout 300 python tools/fuzz_chain.py 11 60 2>&1 | tail -1 > gpurun_out/$R/fuzz_chain.txt
timeout 300 python tools/fuzz_chain.py 12 90 wide 2>&1 | tail -1 > gpurun_out/$R/fuzz_chain_wide.txt
timeout 300 python tools/chain_probe.py 2>&1 | tail -6 > gpurun_out/$R/chain_probe.txt
timeout 300 python tools/fuzz_families.py 31 90 2>&1 | tail -1 > gpurun_out/$R/fuzz_families.txt
timeout 300 python tools/fuzz_ef_packed.py 31 40 2>&1 | tail -1 > gpurun_out/$R/fuzz_ef_packed.txt
timeout 300 python tools/fuzz_graph_roc.py 31 40 2>&1 | tail -1 > gpurun_out/$R/fuzz_graph_roc.txt
timeout 300 python tools/fuzz_wt.py 31 30 2>&1 | tail -1 > gpurun_out/$R/fuzz_wt.txt
timeout 300 python tools/bench_wt.py 2>&1 | tail -4 > gpurun_out/$R/bench_wt.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$R/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
for w in s1_uniform uniform_16m uniform_64m_1k c5 s2_64m; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null >> gpurun_out/$R/bench_other.jsonl
done
for c in packed ef; do for w in s1 uniform_16m uniform_64m_1k; do
  timeout 600 python bench.py --workload $w --codec $c --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null >> gpurun_out/$R/bench_other.jsonl
done; done
# round 6: BASELINE configs[3] evidence (line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE per kernel) + issue / wait counters per kernel
bash tools/prof_graph.sh $R > gpurun_out/$R/prof_graph.log 2>&1
bash tools/pmc_cmd.sh $R graph python tools/bench_graph.py 1000000 64 2 > gpurun_out/$R/pmc_graph.log 2>&1
bash tools/pmc_cmd.sh $R wt python tools/bench_wt.py uniform_16m > gpurun_out/$R/pmc_wt.log 2>&1
bash tools/pmc_cmd.sh $R ef_s2 python bench.py --workload s2 --codec ef --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify > gpurun_out/$R/pmc_ef_s2.log 2>&1
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof_s1 -o s1 -- python bench.py --no-cpu-baseline --no-extra > /dev/null 2> gpurun_out/$R/prof.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof_u16 -o u16 -- python bench.py --workload uniform_16m --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2>> gpurun_out/$R/prof.err
python profiles/extract_rocprof.py gpurun_out/$R/prof_s1/s1_results.db gpurun_out/$R/bench_s1_kernel_stats.csv
python profiles/extract_rocprof.py gpurun_out/$R/prof_u16/u16_results.db gpurun_out/$R/bench_uniform16m_kernel_stats.csv
rm -rf gpurun_out/$R/prof_s1 gpurun_out/$R/prof_u16
timeout 600 python tools/probe_grp.py 2>&1 | tail -10 > gpurun_out/$R/probe_grp.txt
MIN_NS=500000 GPU_MAX_HW_QUEUES=8 bash tools/prof_s2.sh s2 2>&1 | grep "start" | grep -v "^\[" > gpurun_out/$R/s2_timeline.txt
VIDC_SERIAL=1 MIN_NS=500000 GPU_MAX_HW_QUEUES=8 bash tools/prof_s2.sh s2 2>&1 | grep "start" | grep -v "^\[" > gpurun_out/$R/s2_timeline_serial.txt
# S2 step (encode + decode of a fresh object) over 10 rounds, defaults against the round-3 policies (interleaved)
ROUNDS=10 timeout 900 python tools/s2_ab.py - VIDC_B2_MASK=0 VIDC_LANE_ALIGN=4 VIDC_ENC_PRIO=0 VIDC_B2_MASK=0,VIDC_LANE_ALIGN=4,VIDC_ENC_PRIO=0 2>&1 | grep -v amdgpu > gpurun_out/$R/s2_ab.txt
# HBM-side traffic (PMC) of S2 through the three codecs and of the 16 M-id call; kernel stats of the Elias-Fano / packed-bits benches
for c in roc ef packed; do GPU_MAX_HW_QUEUES=8 bash tools/pmc_workload.sh $R s2 $c > /dev/null 2>&1; done
GPU_MAX_HW_QUEUES=8 bash tools/pmc_workload.sh $R uniform_16m roc > /dev/null 2>&1
bash tools/pmc_s1.sh $R > /dev/null 2>&1
bash tools/prof_ef_s2.sh $R > gpurun_out/$R/prof_ef_s2.txt 2>&1
# host-side phases of a 65 536-list call; the register-index construct of DESIGN section 11 outside the library
python tools/trace_host.py uniform_16m 2>&1 | tail -14 > gpurun_out/$R/trace_u16.txt
(hipcc --offload-arch=gfx950 -O2 tools/hw_gpr_idx_probe.hip -o /tmp/probe 2>&1 | tail -3; GPU_MAX_HW_QUEUES=8 timeout 400 /tmp/probe 10 4096 8192 10000 200000 100000) > gpurun_out/$R/hw_gpr_idx_probe.txt 2>&1
cat gpurun_out/$R/pytest_gpu.txt gpurun_out/$R/pytest_r5_old_defaults.txt gpurun_out/$R/pytest_r5_mask_prio.txt gpurun_out/$R/pytest_r5_switches.txt gpurun_out/$R/pytest_pool_poison.txt gpurun_out/$R/search_paths.txt
cat gpurun_out/$R/s2_ab.txt gpurun_out/$R/pytest_lane_loop.txt gpurun_out/$R/pytest_no_length_classes.txt gpurun_out/$R/pytest_force_grp.txt gpurun_out/$R/pytest_wide.txt gpurun_out/$R/pytest_no_lane_pair.txt gpurun_out/$R/s2_repeated_decodes.txt gpurun_out/$R/s2_timeline.txt gpurun_out/$R/probe_grp.txt
cat gpurun_out/$R/pytest_gpu.txt gpurun_out/$R/pytest_force_general.txt gpurun_out/$R/pytest_no_lane.txt gpurun_out/$R/pytest_force_lane.txt gpurun_out/$R/pytest_old_u.txt gpurun_out/$R/pytest_full_prepass_no_lane_reg.txt gpurun_out/$R/fuzz_chain.txt gpurun_out/$R/fuzz_chain_wide.txt gpurun_out/$R/chain_probe.txt gpurun_out/$R/fuzz_families.txt gpurun_out/$R/fuzz_ef_packed.txt gpurun_out/$R/bench_wt.txt gpurun_out/$R/smoke.txt
python - <<PY
import json
for line in open("gpurun_out/$R/bench_other.jsonl"):
    d = json.loads(line)
    print(d["config"]["workload"][:60], d["config"]["codec"], round(d["ms_per_step"], 3), "ms/step", {k: round(v, 3) for k, v in d["kernel_ms"].items()}, round(d["value"] / 1e6, 1), "M IDs/s", "frac", round(d["roofline"]["frac"], 5), "bits", round(d["bits_per_id"], 3), d["verified_roundtrip"])
PY
cat gpurun_out/$R/bench.json
cat gpurun_out/$R/bench_graph.json
