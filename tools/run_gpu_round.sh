b() { tag=$1; shift; envs=$1; shift; env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer "$@" 2>gpurun_out/lane_$tag.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['ms_per_step'])" || tail -3 gpurun_out/lane_$tag.err; }
for n in 0 64 96 128 160 192; do b bf16_cus$n SHM_LANE_CUS=$n --dtype bf16; done
for n in 64 96 128 160; do b bf16_wide4_cus$n "SHM_LANE_CUS=$n SHM_WGRAD_BF16_WIDE=4" --dtype bf16; done
for n in 0 96 128 160 192; do b f32_cus$n SHM_LANE_CUS=$n; done
b bf16_cus0_again SHM_LANE_CUS=0 --dtype bf16
