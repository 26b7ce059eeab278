set -o pipefail
export SHM_TREE_SHA=4d76f4f
bash tools/profile_round.sh r04_b32_bf16 --dtype bf16 --batch 32 || exit 1
bash tools/trace_step.sh r04_b32_bf16 --dtype bf16 --batch 32 || exit 1
mkdir -p gpurun_out/profiles_new && cp profiles/r04_b32_bf16* gpurun_out/profiles_new/
ls gpurun_out/profiles_new | grep b32
