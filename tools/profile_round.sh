#!/bin/bash
# Profiles of a round on the GPU box (TAG=r07 bash tools/profile_round.sh): the r03 script with the round as a parameter (writes under gpurun_out/${TAG}p; the summaries are copied to profiles/ afterwards):
#   step kernel stats, configs[3] kernel stats (fp32 / bf16, one stream), FETCH_SIZE / WRITE_SIZE passes of the step and of
#   configs[3] in both modes (-> traffic JSONs bench.py reads), the calibration passes, and an MFMA-busy pass of both.
#   PARTS="step cfg3 pmc mfma" selects (default: all).
TAG=${TAG:-r07}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}p; mkdir -p $O
cd $R
PARTS=${PARTS:-step cfg3 pmc mfma}
db() { find $O/$1 -name "*results.db" | head -1; }
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has step; then
  rocprofv3 --kernel-trace --stats -d $O/step -o step -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
  python tools/rocpd_stats.py $(db step) 50 > $O/${TAG}_step_kernel_stats.txt
  grep '^{"metric"' $O/step.log | tail -1 > $O/${TAG}_step_bench_line.json
fi
if has cfg3; then
  for m in fp32 bf16; do
    MODE=$m python tools/run_cfg4.py > $O/run_$m.log 2>&1; tail -1 $O/run_$m.log | cut -c1-160
    S2AG_CFG3_STREAMS=1 MODE=$m rocprofv3 --kernel-trace --stats -d $O/cfg3_$m -o cfg3 -- python tools/run_cfg4.py > $O/prof_$m.log 2>&1
    python tools/rocpd_stats.py $(db cfg3_$m) 60 > $O/${TAG}_cfg3_${m}_kernel_stats.txt
    python tools/rocpd_by_grid.py $(db cfg3_$m) _k > $O/${TAG}_cfg3_${m}_by_grid.txt
  done
fi
if has pmc; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cf -o cf -- python tools/pmc_calib.py > $O/cf.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cw -o cw -- python tools/pmc_calib.py > $O/cw.log 2>&1
  python tools/rocpd_pmc.py $(db cf) > $O/${TAG}_pmc_calib_FETCH_SIZE.txt
  python tools/rocpd_pmc.py $(db cw) > $O/${TAG}_pmc_calib_WRITE_SIZE.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > $O/pf.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > $O/pw.log 2>&1
  python tools/rocpd_pmc.py $(db pf) > $O/${TAG}_pmc_FETCH_SIZE.txt
  python tools/rocpd_pmc.py $(db pw) > $O/${TAG}_pmc_WRITE_SIZE.txt
  python tools/pmc_traffic.py $(db pf) $(db pw) $(db cf) $(db cw) > $O/${TAG}_pmc_traffic.json
  for m in fp32 bf16; do
    S2AG_CFG3_STREAMS=1 S2AG_CFG3_EAGER=1 ITERS=3 MODE=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c3f_$m -o c3f -- python tools/run_cfg4.py > $O/c3f_$m.log 2>&1
    S2AG_CFG3_STREAMS=1 S2AG_CFG3_EAGER=1 ITERS=3 MODE=$m rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c3w_$m -o c3w -- python tools/run_cfg4.py > $O/c3w_$m.log 2>&1
    python tools/rocpd_pmc.py $(db c3f_$m) > $O/${TAG}_pmc_cfg3_${m}_FETCH_SIZE.txt
    python tools/rocpd_pmc.py $(db c3w_$m) > $O/${TAG}_pmc_cfg3_${m}_WRITE_SIZE.txt
    python tools/pmc_traffic.py $(db c3f_$m) $(db c3w_$m) $(db cf) $(db cw) --per-iteration $(grep -o 'iterations_run=[0-9]*' $O/c3f_$m.log | tail -1 | cut -d= -f2) > $O/${TAG}_pmc_traffic_cfg3_$m.json
  done
fi
if has mfma; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/mb -o mb -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > $O/mb.log 2>&1
  python tools/rocpd_pmc.py $(db mb) > $O/${TAG}_pmc_step_MFMA_BUSY.txt
  for m in fp32 bf16; do
    S2AG_CFG3_STREAMS=1 S2AG_CFG3_EAGER=1 ITERS=3 MODE=$m rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/c3m_$m -o c3m -- python tools/run_cfg4.py > $O/c3m_$m.log 2>&1
    python tools/rocpd_pmc.py $(db c3m_$m) > $O/${TAG}_pmc_cfg3_${m}_MFMA_BUSY.txt
  done
fi
for m in step cfg3_fp32 cfg3_bf16; do [ -f $O/${TAG}_pmc_${m}_MFMA_BUSY.txt ] && python tools/mfma_util.py $O/${TAG}_pmc_${m}_MFMA_BUSY.txt > $O/${TAG}_mfma_util_${m}.txt; done
find $O -name "*.db" -delete
ls -la $O | head -60
