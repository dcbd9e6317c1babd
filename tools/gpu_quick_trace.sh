#!/bin/bash
# kernel trace of one python command, summarised per (kernel, grid): tools/gpu_quick_trace.sh <tag> <script> [args]
tag=$1; shift
out=$PWD/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/qt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qt -- python "$@" > $out/${tag}.log 2>$out/${tag}.err
kt=$(find /tmp/qt -name '*kernel_trace.csv' | head -1)
python tools/trace_by_shape.py $kt $out/${tag}_by_shape.json > $out/${tag}_by_shape.txt
tail -3 $out/${tag}.log; head -12 $out/${tag}_by_shape.txt
