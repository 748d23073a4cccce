# round 4, first probe: host trace of the 65 536 x 256 call, S2 timelines (as shipped / one class after the other)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/trace_u16.py > gpurun_out/r04a/trace_u16.txt 2>&1
MIN_NS=300000 GPU_MAX_HW_QUEUES=8 bash tools/prof_s2.sh s2 2>&1 | grep "start" | grep -v "^\[" > gpurun_out/r04a/s2_timeline.txt
VIDC_SERIAL=1 MIN_NS=300000 GPU_MAX_HW_QUEUES=8 bash tools/prof_s2.sh s2 2>&1 | grep "start" | grep -v "^\[" > gpurun_out/r04a/s2_timeline_serial.txt
cat gpurun_out/r04a/trace_u16.txt | tail -60
