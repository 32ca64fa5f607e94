#!/bin/bash
# Regenerates what profiles/ holds (run on the GPU box through gpurun; outputs under gpurun_out/prof_refresh, copied into
# profiles/ as r<NN>_* afterwards).  rocprofv3 gets the program itself after "--" (python3 script), counters in their own passes.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_refresh
rm -rf $O && mkdir -p $O
python3 bench.py --steps 50 --warmup 10 --breakdown > $O/bench.json 2> $O/bench_breakdown.txt
LO_PROF_LAYERS=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --config2-steps 0 --highend-steps 0 --breakdown > /dev/null 2> $O/bench_per_layer.txt
python3 tools/vendor_ceiling.py --mine $O/bench_per_layer.txt > $O/vendor_ceiling.txt 2> $O/vendor_ceiling.err
python3 tools/vendor_teacher_conv.py > $O/vendor_teacher_conv.txt 2> $O/vendor_teacher_conv.err
python3 tools/fp8_layer_table.py > $O/fp8_per_layer.txt 2> $O/fp8_per_layer.err
python3 tools/dp_overhead_probe.py > $O/dp_path_probe.txt 2>&1
python3 tools/hybrid_probe.py > $O/hybrid_breakdown.txt 2>&1
python3 tools/gnb_det.py 8 > $O/gnb_det.log 2>&1
python3 tools/host_enqueue_probe.py > $O/host_enqueue.txt 2>&1
for cfg in "64 8 512" "16 32 128" "8 64 64"; do set -- $cfg; python3 tools/op_bench.py --op attn --B $1 --H $2 --Cin $3 2>&1 | grep attn >> $O/selfattn2d_op_bench.txt; done
echo "bench + probes done" >&2
# kernel durations are compared with the side stream switched off in BOTH measurements (the library's HIP-event leg always runs
# that way): under concurrency - and under the tracer's kernel serialisation of a two-stream schedule - a kernel's duration is not a
# property of the kernel.  The default schedule is traced as well (..._overlap_on).
FAST="--no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --config2-steps 0 --highend-steps 0"
export LO_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench -- python3 bench.py --steps 20 --warmup 5 $FAST > $O/bench_under_rocprof.json 2> $O/rp_bench.err
unset LO_NO_OVERLAP
echo "trace 1 done" >&2
rocprofv3 --kernel-trace --stats -d $O/rp_bench_on -o bench -- python3 bench.py --steps 20 --warmup 5 $FAST > $O/bench_under_rocprof_overlap_on.json 2> $O/rp_bench_on.err
echo "trace 2 done" >&2
rocprofv3 --kernel-trace --stats -d $O/rp_hybrid -o hybrid -- python3 tools/hybrid_probe.py > $O/hybrid_under_rocprof.txt 2> $O/rp_hybrid.err
echo "trace 3 done" >&2
rocprofv3 --kernel-trace --stats -d $O/rp_fullbwd -o fb -- python3 tools/full_backward_probe.py > $O/full_backward_probe.txt 2> $O/rp_fullbwd.err
echo "trace 4 done" >&2
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --min-warmup 20 $FAST --prof-steps 0 > /dev/null 2> $O/pmc_fetch.err
echo "pmc 1 done" >&2
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --min-warmup 20 $FAST --prof-steps 0 > /dev/null 2> $O/pmc_write.err
echo "pmc 2 done" >&2
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_h -o f -- python3 tools/hybrid_probe.py > /dev/null 2> $O/pmc_fetch_h.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_h -o w -- python3 tools/hybrid_probe.py > /dev/null 2> $O/pmc_write_h.err
echo "pmc hybrid done" >&2
python3 tools/rocpd_extract.py stats $(find $O/rp_bench -name "*results.db" | head -1) $O/bench_kernel_stats.csv
python3 tools/rocpd_extract.py stats $(find $O/rp_bench_on -name "*results.db" | head -1) $O/bench_kernel_stats_overlap_on.csv
python3 tools/rocpd_extract.py stats $(find $O/rp_hybrid -name "*results.db" | head -1) $O/hybrid_kernel_stats.csv
python3 tools/rocpd_extract.py stats $(find $O/rp_fullbwd -name "*results.db" | head -1) $O/teacher_full_backward_kernel_stats.csv
python3 tools/rocpd_extract.py traffic $(find $O/pmc_fetch -name "*results.db" | head -1) $(find $O/pmc_write -name "*results.db" | head -1) $O/traffic.json
python3 tools/rocpd_extract.py traffic $(find $O/pmc_fetch_h -name "*results.db" | head -1) $(find $O/pmc_write_h -name "*results.db" | head -1) $O/traffic_hybrid.json
# keep the merge-back small: drop the raw traces, keep stats + counter csv
find $O -name "*kernel_trace*" -delete
find $O -name "*.db" -delete
du -sh $O
