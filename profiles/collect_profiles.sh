#!/bin/bash
# usage: collect_profiles.sh <tag> cfg...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
for c in "$@"; do
  B="python $R/bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $c -- $B > $R/gpurun_out/prof_${tag}_$c.log 2>&1
  grep -h "^{\"metric" $R/gpurun_out/prof_${tag}_$c.log | tail -1 > $R/gpurun_out/prof_$tag/${c}_bench.json
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -o $c -- $B --no-graph > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -o $c -- $B --no-graph > /dev/null 2>&1
  rm -f $R/gpurun_out/prof_$tag/${c}_kernel_trace.csv $R/gpurun_out/prof_${tag}_fetch/${c}_kernel_trace.csv $R/gpurun_out/prof_${tag}_write/${c}_kernel_trace.csv
  echo "$c done: $(head -2 $R/gpurun_out/prof_$tag/${c}_kernel_stats.csv | tail -1 | cut -c1-120)"
done
