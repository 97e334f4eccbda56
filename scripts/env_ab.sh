run() { echo "$1 rot $(env $1 timeout 300 python bench.py --no-cpu-baseline --no-decode --steps 60 --timed-only 2>/dev/null | grep -o '"ms_per_step": [0-9.]*') sta $(env $1 timeout 300 python bench.py --no-cpu-baseline --no-decode --steps 60 --timed-only --static-batch 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for i in 1 2; do
  run X=0
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run HSA_ENABLE_INTERRUPT=0
  run HSA_ENABLE_SDMA=0
done
