timeout -k 10 200 python tools/probes/bwd_blocks.py reduce 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/probes/bwd_blocks.py apply 2>&1 | grep -v amdgpu
for rb in 0 768 1024 2048; do
  SHM_ELEM_REDUCE_BLOCKS=$rb timeout -k 10 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reduce_blocks $rb', j['ms_per_step'])"
done
