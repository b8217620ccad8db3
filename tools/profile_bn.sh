export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bnp; mkdir -p $O; cd $R
for v in 0 1; do
S2AG_BN_FUSED=$v rocprofv3 --kernel-trace --stats -d $O/f$v -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/f$v.log 2>&1
python tools/rocpd_stats.py $(find $O/f$v -name "*results.db" | head -1) 60 | grep -i "bn_\|dispatches" > $O/bn_$v.txt
done
find $O -name "*.db" -delete
