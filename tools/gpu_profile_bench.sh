#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel trace of the same command summarised per (kernel, grid),
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE separately, no tracing flags beside the counter) over tools/k4_headline.py.
# Usage: tools/gpu_profile_bench.sh <tag>      -> gpurun_out/<tag>_*
set -e
tag=${1:-r02}
out=$PWD/gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -3 $out/${tag}_bench.err
export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python bench.py > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
kt=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
ks=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
python tools/trace_by_shape.py $kt $out/${tag}_kernel_by_shape.json > $out/${tag}_kernel_by_shape.txt
cp $ks $out/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python tools/k4_headline.py 40 > $out/${tag}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python tools/k4_headline.py 40 > $out/${tag}_pmc_w.log 2>&1
python tools/pmc_summarise.py /tmp/prof_f /tmp/prof_w $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_traffic.txt
cp $(find /tmp/prof_f -name '*counter_collection.csv' | head -1) $out/${tag}_pmc_fetch_counter_collection.csv
cp $(find /tmp/prof_w -name '*counter_collection.csv' | head -1) $out/${tag}_pmc_write_counter_collection.csv
echo done
