#!/bin/bash
# usage (on the GPU box, through gpurun): bash profiles/collect_profiles.sh <tag> cfg...
# per config: a --kernel-trace --stats pass, two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with
# other tracing) of the same bench.py command, then a clean bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
mkdir -p $R/gpurun_out/bench_$tag
for c in "$@"; do
  B="python $R/bench.py --config $c --steps 50 --warmup 5 --repeats 5 --configs none --no-cpu-baseline --no-variants --no-verify"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $c -- $B > $R/gpurun_out/prof_${tag}_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -o $c -- $B --no-graph > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -o $c -- $B --no-graph > /dev/null 2>&1
  rm -f $R/gpurun_out/prof_$tag/${c}_kernel_trace.csv $R/gpurun_out/prof_${tag}_fetch/${c}_kernel_trace.csv $R/gpurun_out/prof_${tag}_write/${c}_kernel_trace.csv
  # keep the counter files small enough to travel: only our kernels
  for k in fetch write; do
    f=$R/gpurun_out/prof_${tag}_$k/${c}_counter_collection.csv
    [ -f $f ] && (head -1 $f; grep -E "k_transition|k_rolling|k_mask_step|k_env_step|k_episode|k_macs|k_dyn_bits|k_big" $f) > $f.tmp && mv $f.tmp $f
  done
  # the clean, fully reported bench line comes LAST: traffic.json is refreshed from the passes above first, so the line's
  # roofline.traffic / kernel_us_rocprof are this session's
  (cd $R && python profiles/summarize.py $tag $c > /dev/null 2>&1)
  timeout 400 python $R/bench.py --config $c --configs none > $R/gpurun_out/bench_$tag/$c.json 2> $R/gpurun_out/bench_$tag/$c.err
  echo "$c done: $(head -c 300 $R/gpurun_out/bench_$tag/$c.json)"
done
