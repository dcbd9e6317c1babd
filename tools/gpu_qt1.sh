#!/bin/bash
# The record gpurun_out/r2_qt1.log, re-run with variants (DESIGN.md "r2_qt1"): each line is one rocprofv3 --kernel-trace run of
# tools/probe_qt1.py; a host-side SIGSEGV of the profiled process is the expected outcome of the `nokeep` ones.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { tag=$1; shift; ( for kv in "$@"; do export "$kv"; done; rocprofv3 --kernel-trace -d gpurun_out/qt1_$tag -- python3 tools/probe_qt1.py > gpurun_out/r3_qt1_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/r3_qt1_$tag.log ); }
run keep LPX_FUSED_PIVOT=0
run keep3 LPX_FUSED_PIVOT=0 QT1_CAPS=2000,10000,3000,2000,10000
run nokeep LPX_FUSED_PIVOT=0 LPX_GRAPH_KEEP=0
run nokeep_mid LPX_FUSED_PIVOT=0 LPX_GRAPH_KEEP=0 QT1_SHAPE=2048,4096
echo done
