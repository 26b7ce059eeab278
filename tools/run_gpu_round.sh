set -o pipefail
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-shape gpurun_out/r4_pershape_f32.json > gpurun_out/r4_ps_f32.json 2> gpurun_out/r4_ps_f32.err; echo "f32 rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --per-shape gpurun_out/r4_pershape_bf16.json > gpurun_out/r4_ps_bf16.json 2> gpurun_out/r4_ps_bf16.err; echo "bf16 rc=$?"
