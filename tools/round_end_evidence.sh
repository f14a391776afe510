#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun from the repo root): PMC traffic of the real step per launch shape,
# rocprofv3 kernel traces of bench.py per configuration, one critic update kernel by kernel.  usage: tools/round_end_evidence.sh <tag>
# Every rocprofv3 call has the program itself behind `--`; counters are collected in passes of their own.
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "64 msvd" "128 msvd" "64 msrvtt"; do
  set -- $cfg; B=$1; S=$2; N=${S}_b$B
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$N -- python3 $R/tools/pmc_step_target.py $O/calls_$N.json $B $S > $O/pmc_fetch_$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$N -- python3 $R/tools/pmc_step_target.py - $B $S > $O/pmc_write_$N.log 2>&1
  F=$(find $O/pmc_fetch_$N -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write_$N -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_step_traffic.py $F $W $O/calls_$N.json > $O/traffic_$N.json 2> $O/traffic_$N.err
  rm -rf $O/pmc_fetch_$N $O/pmc_write_$N
  EXTRA="--no-pass --no-cpu-baseline --no-eager-baseline --no-batch128 --no-gan --no-inference --no-msrvtt --no-sustained --no-dp-schedule"
  rocprofv3 --kernel-trace --stats -d $O/trace_$N -o t -- python3 $R/bench.py --batch $B --shape $S --steps 20 --warmup 5 $EXTRA > $O/bench_line_under_rocprof_$N.json 2> $O/trace_$N.log
  D=$(find $O/trace_$N -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $D 60 1 > $O/kernel_stats_bench_$N.csv
  rm -rf $O/trace_$N
done
rocprofv3 --kernel-trace --stats -d $O/trace_critic -o t -- python3 $R/tools/critic_profile.py 64 10 5 > $O/critic_profile.log 2>&1
D=$(find $O/trace_critic -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $D 80 60 > $O/kernel_stats_critic_update.csv     # (10 + 2 calls) x 5 updates
rm -rf $O/trace_critic
cd $R && python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/bench_full.json
