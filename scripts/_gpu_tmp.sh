export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sync_ln.py -m gpu -q -p no:cacheprovider -x -k "projection_inside or attention_inside" > gpurun_out/t_proj.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_proj.log
ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_trace.so timeout 300 python scripts/attn_out_ln_trace.py 2>&1 | grep -v "amdgpu.ids\|XCD" | head -17
for rep in 1 2 3; do
for v in old new; do
lib=$PWD/zero_amd/csrc/libzero_hip_old.so; [ $v = new ] && lib=$PWD/zero_amd/csrc/libzero_hip.so
echo -n "$v: "; ZERO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 400 --warmup 20 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
