export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2 3 4 5; do
for v in def pre; do
lib=$PWD/zero_amd/csrc/libzero_hip_pre.so; [ $v = def ] && lib=$PWD/zero_amd/csrc/libzero_hip.so
echo -n "$v: "; ZERO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 600 --warmup 30 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
