# long differential fuzz of the final kernels (all ROC kernel-family modes incl. row-per-list and lane pairs, chain kernels, Elias-Fano /
# packed bits) + the repeated-decode stress on S2 and the half-size S2 -> gpurun_out/<tag>/long_fuzz.txt     usage: bash tools/long_fuzz.sh <tag> [seconds per fuzzer] [seed base]
cd $GRAFT_REPO_ROOT; R=${1:-r03g}; T=${2:-420}; S=${3:-300}; mkdir -p gpurun_out/$R
(
timeout $((T+60)) python tools/fuzz_families.py $((S+1)) $T 2>&1 | tail -1
GPU_MAX_HW_QUEUES=8 timeout $((T+60)) python tools/fuzz_families.py $((S+2)) $T 2>&1 | tail -1
timeout $((T/2+60)) python tools/fuzz_chain.py $((S+3)) $((T/2)) 2>&1 | tail -1
timeout $((T/2+60)) python tools/fuzz_chain.py $((S+4)) $((T/2)) wide 2>&1 | tail -1
timeout $((T/2+60)) python tools/fuzz_ef_packed.py $((S+5)) $((T/2)) 2>&1 | tail -1
echo "S2 decoded 400 times (8 hardware queues, 4 / 8 / 5 / 6 streams), compared with the first decode:"
GPU_MAX_HW_QUEUES=8 NQS=4,8,5,6 ITERS=400 timeout 300 python tools/repro_s2b.py "X=1" 2>&1 | grep "differs\|it 399" | cut -c1-160
echo "S2 decoded 400 times (4 hardware queues, 3 streams):"
GPU_MAX_HW_QUEUES=4 NQS=3 ITERS=400 timeout 300 python tools/repro_s2b.py "X=1" 2>&1 | grep "differs\|it 399" | cut -c1-160
echo "half-list-size S2 (5e8 ids in 2^21 lists: the one-lane register decoder in the mix) decoded 400 times:"
GPU_MAX_HW_QUEUES=8 ZIPF=500000000:2097152 NQS=6,8 ITERS=400 timeout 300 python tools/repro_s2b.py "X=1" 2>&1 | grep "differs\|it 399" | cut -c1-160
) > gpurun_out/$R/long_fuzz.txt 2>&1
cat gpurun_out/$R/long_fuzz.txt
