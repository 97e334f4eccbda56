export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2 3; do
for v in old new; do
lib=$PWD/zero_amd/csrc/libzero_hip_old.so; [ $v = new ] && lib=$PWD/zero_amd/csrc/libzero_hip.so
echo -n "$v: "; ZERO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 400 --warmup 20 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*\|"gnorm": [0-9.]*' | tr '\n' ' '; echo
done
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sync_ln.py -m gpu -q -p no:cacheprovider -x -k "drop or attention" 2>&1 | tail -3
