cd /tmp
for mode in rot sta; do
  extra=""; [ $mode = sta ] && extra="--static-batch"
  rm -rf $OLDPWD/gpurun_out/pf_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/pf_$mode -o r1 -- python $OLDPWD/bench.py --steps 40 --warmup 2 --timed-only $extra > $OLDPWD/gpurun_out/pf_$mode.log 2>&1
  python $OLDPWD/scripts/prof_summary.py $(find $OLDPWD/gpurun_out/pf_$mode -name "*.db" | head -1) 42 > $OLDPWD/gpurun_out/pf_$mode.txt 2>&1
  python $OLDPWD/scripts/prof_gaps.py $(find $OLDPWD/gpurun_out/pf_$mode -name "*.db" | head -1) > $OLDPWD/gpurun_out/pf_gaps_$mode.txt 2>&1
  rm -rf $OLDPWD/gpurun_out/pf_$mode
done
cd $OLDPWD
head -8 gpurun_out/pf_gaps_rot.txt; head -8 gpurun_out/pf_gaps_sta.txt
