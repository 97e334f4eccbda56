export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2; do
for t in 0 1 2 4 7; do
echo -n "tune19=$t: "; ZERO_HIP_TUNE=19:$t timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 400 --warmup 20 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
