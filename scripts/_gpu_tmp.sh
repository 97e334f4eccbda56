export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2 3; do
for v in def Os O2; do
lib=$PWD/zero_amd/csrc/libzero_hip_$v.so; [ $v = def ] && lib=$PWD/zero_amd/csrc/libzero_hip.so
echo -n "$v: "; ZERO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 400 --warmup 20 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
