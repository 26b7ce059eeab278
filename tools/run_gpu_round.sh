set -o pipefail
timeout -k 10 1150 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r4_full.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_full.log | tail -12
timeout -k 10 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
