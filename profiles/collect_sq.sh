#!/bin/bash
# usage: [BENCH_FLAGS="..."] collect_sq.sh <tag> <config> <name> COUNTER...   (one rocprofv3 --pmc pass, kernel trace only; run through gpurun)
# writes gpurun_out/prof_<tag>_sq/<config>_<name>_counter_collection.csv; profiles/summarize_sq.py condenses it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; cfg=$2; name=$3; shift 3
B="python $R/bench.py --config $cfg --steps 20 --warmup 3 --repeats 1 --configs none --no-cpu-baseline --no-variants --no-verify --no-graph $BENCH_FLAGS"
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof_${tag}_sq -o ${cfg}_${name} -- $B > $R/gpurun_out/prof_${tag}_sq_${cfg}_${name}.log 2>&1
rm -f $R/gpurun_out/prof_${tag}_sq/${cfg}_${name}_kernel_trace.csv
# keep the counter file small enough to travel: only our kernels
f=$R/gpurun_out/prof_${tag}_sq/${cfg}_${name}_counter_collection.csv
[ -f $f ] && (head -1 $f; grep -E "k_transition|k_rolling|k_mask_step|k_env_step|k_episode|k_macs|k_dyn_bits|k_big" $f) > $f.tmp && mv $f.tmp $f
ls -la $R/gpurun_out/prof_${tag}_sq/ | tail -5
