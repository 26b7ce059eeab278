set -o pipefail
for v in base w h base w h; do
  L=""; [ $v != base ] && L=$PWD/build_ab/lib_$v.so
  SHM_LIB_PATH=$L timeout -k 10 200 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_ab_$v.json 2>gpurun_out/r4_ab_$v.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ab_$v.json") if l.startswith("{")][0])
k=j["roofline"]["kernels"]
print("bf16 $v", j["ms_per_step"], j["value"], "north*", j["north_star_block"]["us"], "wreg16<2>", k["tapgemm_wreg16_bf16_kernel<2>"]["ms_per_step"], "halo128", k["tapgemm_halo_kernel<__bf16, __bf16, 128, 16, true, 2>"]["ms_per_step"], "halo64", k["tapgemm_halo_kernel<__bf16, __bf16, 64, 16, true, 2>"]["ms_per_step"])
PY
done
