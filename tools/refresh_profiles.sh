#!/bin/bash
# Regenerates what profiles/ holds (run on the GPU box through gpurun; outputs under gpurun_out/prof_refresh).
# rocprofv3 gets the program itself after "--" (python3 script), counters in their own passes.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_refresh
rm -rf $O && mkdir -p $O
python3 bench.py --steps 50 --warmup 10 --breakdown > $O/bench.json 2> $O/bench_breakdown.txt
python3 tools/hybrid_probe.py > $O/hybrid_breakdown.txt 2>&1
# kernel durations are compared with the side stream switched off in BOTH measurements (the library's HIP-event leg always runs
# that way): under concurrency - and under the tracer's kernel serialisation of a two-stream schedule - a kernel's duration is not a
# property of the kernel.  The default schedule is traced as well (..._overlap_on).
export LO_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 > $O/bench_under_rocprof.json 2> $O/rp_bench.err
unset LO_NO_OVERLAP
rocprofv3 --kernel-trace --stats -d $O/rp_bench_on -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 > $O/bench_under_rocprof_overlap_on.json 2> $O/rp_bench_on.err
rocprofv3 --kernel-trace --stats -d $O/rp_hybrid -o hybrid -- python3 tools/hybrid_probe.py > $O/hybrid_under_rocprof.txt 2> $O/rp_hybrid.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --prof-steps 0 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --prof-steps 0 > /dev/null 2> $O/pmc_write.err
python3 tools/rocpd_extract.py stats $O/rp_bench/bench_results.db $O/bench_kernel_stats.csv
python3 tools/rocpd_extract.py stats $O/rp_bench_on/bench_results.db $O/bench_kernel_stats_overlap_on.csv
python3 tools/rocpd_extract.py stats $O/rp_hybrid/hybrid_results.db $O/hybrid_kernel_stats.csv
python3 tools/rocpd_extract.py traffic $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db $O/traffic.json
find $O -name "*stats*" | head -20
# keep the merge-back small: drop the raw traces, keep stats + counter csv
find $O -name "*kernel_trace*" -delete
find $O -name "*.db" -size +20M -delete
du -sh $O
