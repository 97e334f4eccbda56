export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sync_ln.py -m gpu -q -p no:cacheprovider -x -k "projection_inside" > gpurun_out/t_proj.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/t_proj.log
for m in 1 0 1 0; do
ZERO_HIP_PROJ_ATTN=$m timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 40 > gpurun_out/benchq_proj$m.json 2> gpurun_out/benchq_proj$m.err; echo "PROJ_ATTN=$m rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*\|"launches_per_step": [0-9]*' gpurun_out/benchq_proj$m.json | head -3; tail -2 gpurun_out/benchq_proj$m.err
done
