set -o pipefail
timeout -k 10 900 python -m pytest tests/test_variants_gpu.py tests/test_train_loop_gpu.py tests/test_bf16_gpu.py -m gpu -q -x -k "wgrad or reproducible or bf16_train_step or config3" > gpurun_out/r4_t9.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_t9.log | tail -12
if [ $rc -eq 0 ]; then
timeout -k 10 300 python tools/bench_wgrad_bf16.py 96,16,512,1024,2 96,256,32,64,2 48,32,512,1024,2 2>&1 | grep -v amdgpu
for w in 0 1; do SHM_WGRAD_BF16_WIDE=$w timeout -k 10 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide $w', j['ms_per_step'])"; done
fi
