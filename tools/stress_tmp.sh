#!/bin/bash
N=${N:-10}; bad=0; mkdir -p gpurun_out/stress
for i in $(seq 1 $N); do
  python -X faulthandler bench.py --no-cpu-baseline > /tmp/st.log 2>&1 || { bad=$((bad+1)); cp /tmp/st.log gpurun_out/stress/crash_$i.log; grep -m3 "Assertion\|Segmentation\|Aborted\|Fatal" /tmp/st.log | cut -c1-160; }
done
echo "CFG[$TAG] crashes: $bad / $N"
