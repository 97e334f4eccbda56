export PYTHONPATH=$PWD TMPDIR=/tmp
for t in 0 1; do
echo "##### trace tune 17:$t"
ZERO_HIP_TUNE=17:$t ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_trace.so timeout 300 python scripts/attn_out_ln_trace.py 2>&1 | grep -v "half step [2-9]\|amdgpu.ids"
done
for t in 0 1 7 0 1 7; do
ZERO_HIP_TUNE=17:$t timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 40 > gpurun_out/benchq_t$t.json 2> gpurun_out/benchq_t$t.err; echo "TUNE17=$t rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*' gpurun_out/benchq_t$t.json | head -3 | tr '\n' ' '; echo
done
