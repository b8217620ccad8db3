#!/bin/bash
# r05 call A: the full -m gpu suite at HEAD (no -x), then one un-profiled default bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/t_all.log 2>&1; echo "full suite rc=$?"; tail -40 $O/t_all.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-600 $O/bench_line.json; tail -5 $O/bench.log | cut -c1-300
