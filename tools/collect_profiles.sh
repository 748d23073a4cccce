# copies the evidence of a tools/final_run.sh run from gpurun_out/<tag>/ into profiles/ (tracked): usage: bash tools/collect_profiles.sh <tag>
R=${1:-r04a}; S=gpurun_out/$R; D=profiles
cp $S/bench.json $D/${R}_bench_s1.json
cp $S/bench_s1_kernel_stats.csv $D/${R}_bench_s1_kernel_stats.csv
cp $S/bench_uniform16m_kernel_stats.csv $D/${R}_bench_uniform16m_kernel_stats.csv
cp $S/bench_s1_pmc_fetch.csv $D/${R}_bench_s1_pmc_fetch.csv
cp $S/bench_s1_pmc_write.csv $D/${R}_bench_s1_pmc_write.csv
cp $S/pmc_traffic.json $D/pmc_traffic_s1.json
for t in s2_roc s2_ef s2_packed uniform_16m_roc; do cp $S/pmc_traffic_$t.json $D/pmc_traffic_$t.json; cp $S/pmc_traffic_$t.json $D/${R}_pmc_traffic_$t.json; done
cp $S/s2_timeline.txt $D/${R}_s2_timeline.txt
cp $S/s2_timeline_serial.txt $D/${R}_s2_timeline_serial.txt
cp $S/s2_ab.txt $D/${R}_s2_ab.txt
cp $S/s2_repeated_decodes.txt $D/${R}_s2_repeated_decodes.txt
cp $S/bench_other.jsonl $D/${R}_bench_other_workloads.jsonl
cp $S/bench_graph.json $D/${R}_bench_graph_1M_x_64.json
cp $S/probe_grp.txt $D/${R}_probe_grp.txt
for c in ef packed; do for w in s2 uniform_16m s1; do cp $S/${c}_${w}_kernel_stats.csv $D/${R}_bench_${c}_${w}_kernel_stats.csv; done; done
(for f in pytest_gpu pytest_force_general pytest_no_lane pytest_force_lane pytest_old_u pytest_full_prepass_no_lane_reg pytest_force_grp pytest_wide pytest_no_lane_pair pytest_lane_loop pytest_no_length_classes pytest_no_avx2 pytest_r5_switches pytest_r5_old_defaults pytest_r5_mask_prio pytest_pool_poison; do echo "== $f"; cat $S/$f.txt; done
 for f in fuzz_chain fuzz_chain_wide fuzz_families fuzz_ef_packed fuzz_graph_roc fuzz_wt chain_probe bench_wt smoke; do echo "== $f"; cat $S/$f.txt; done) | grep -v "amdgpu.ids" > $D/${R}_tests_all_modes.txt
[ -f $S/search_paths.txt ] && cp $S/search_paths.txt $D/${R}_search_paths.txt
[ -f $S/trace_u16.txt ] && grep -v amdgpu.ids $S/trace_u16.txt > $D/${R}_trace_uniform16m_host.txt
[ -f $S/hw_gpr_idx_probe.txt ] && cp $S/hw_gpr_idx_probe.txt $D/${R}_hw_gpr_idx_probe.txt
for f in bench_graph_kernel_stats.csv pmc_traffic_graph_ef.json pmc_traffic_graph_compact.json pmc_traffic_graph_roc.json pmc_graph.json pmc_wt.json pmc_ef_s2.json; do [ -f $S/$f ] && cp $S/$f $D/${R}_$f; done
for f in pmc_issue_s1_roc.json pmc_issue_s2_roc.json; do [ -f $S/$f ] && cp $S/$f $D/$f && cp $S/$f $D/${R}_$f; done
ls $D | grep "^${R}_" | wc -l
