"""bench.py's N > 1 path as the driver will run it, rehearsed on the one GPU of the test box: `python bench.py --gpus 2` becomes a
launcher (no GPU call in it) that starts the two ranks through torch.distributed.run -- the command line the driver uses for
--gpus N -- with `--same-device --backend gloo` (RCCL refuses two ranks on one device).  Asserts the JSON contract of the N > 1
line: whole-job value, the ranks that really ran, and the `comm` object (collective time on the reducer stream, bytes, and the
exposed part the main stream waited for) that makes a scaling number explainable."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_two_ranks_on_one_gpu_through_the_launcher():
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 alone prints, one line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"]["world_size"] == 2 and j["rccl_ranks"]["backend"] == "gloo"
    assert j["rccl_ranks"]["same_device"] is True and j["scaling"] == "weak" and j["config"]["global_batch"] == 16
    assert j["config"]["losses_finite"] is True and "cpu_baseline" not in j
    assert abs(j["value"] - 16 / (j["ms_per_step"] * 1e-3)) < 1e-2 * j["value"]          # whole-job images/s over both ranks
    c = j["comm"]
    # one D bucket + five G buckets per step: the two flat fp32 gradients, 26.4 + 74.1 MB at S=256
    assert c["collectives"] == 6 and c["bytes"] == 4 * (6605504 + 18525569)
    assert c["d_bucket_ms"] > 0 and c["g_buckets_ms"] > 0 and 0 <= c["exposed_ms"] <= j["ms_per_step"]
    # the per-kernel replay ran on rank 0 under N = 2 as well (its durations are not asserted: under this host-staged rehearsal
    # backend an event bracket also absorbs collective stalls)
    assert j["roofline"]["kernel"] and j["roofline"]["launches"] > 0
