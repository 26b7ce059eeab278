"""world_size-2 data-parallel path on CPU (gloo): the reducer sums flat buckets across ranks, and
rank-local B=1 gradients averaged over 2 ranks equal one B=2 step under the batch rule
(SURVEY 8(e)); the oracle stands in for the device step (no GPU here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import step_torch as st
    from shmgan_amd.dist import GradReducer, world_size
    assert world_size() == world
    S, F = 64, 16
    g, d, gb, db = st.init_params(F, S)
    inp2 = st.make_inputs(2, S)
    dr = st.make_draws(1, 2, S, F)
    sf = st.style_factor_intended(S)
    drb = st.StepDraws(dr.flags, dr.target_label, dr.noise[[rank, 2 + rank]], dr.keep_mask[[rank, 2 + rank]])
    r = st.train_step(g, d, gb, db, [a[rank:rank + 1] for a in inp2], drb, sf, F)
    red = GradReducer()
    flat_d = torch.cat([t.reshape(-1) for t in r["gD"]])
    flat_g = torch.cat([t.reshape(-1) for t in r["gG"]])
    ev = red.allreduce_async(flat_d)
    red.wait(ev)
    red.allreduce_async(flat_g)
    if rank == 0:
        full = st.train_step(g, d, gb, db, inp2, dr, sf, F)
        ref_d = torch.cat([t.reshape(-1) for t in full["gD"]])
        ref_g = torch.cat([t.reshape(-1) for t in full["gG"]])
        q.put((float((flat_d / world - ref_d).abs().max() / ref_d.abs().max()),
               float((flat_g / world - ref_g).abs().max() / ref_g.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_batch_of_two():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ed, eg = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ed < 1e-9 and eg < 1e-9, (ed, eg)


def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from shmgan_amd.dist import GradExchangeError, GradReducer, init_process_group
    init_process_group("gloo", timeout_s=5, rank=rank, world_size=world)
    red = GradReducer()
    ok = torch.ones(4)
    red.allreduce_async(ok)                       # a healthy exchange first
    assert float(ok[0]) == world
    if rank == 1:
        os._exit(0)                               # the peer dies between two steps
    try:
        red.allreduce_async(torch.ones(1 << 16))
        q.put("no error")
    except GradExchangeError as e:
        q.put("GradExchangeError: " + str(e)[:80])
        raise SystemExit(3)                       # what a training script does: non-zero exit, no retry


def test_a_dead_rank_fails_the_exchange_instead_of_hanging():
    """GradReducer's failure path: the surviving rank gets GradExchangeError within the process-group timeout
    (shmgan_amd.dist.init_process_group sets a finite one) and exits non-zero."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msg = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert msg.startswith("GradExchangeError"), msg
    assert procs[0].exitcode == 3 and procs[1].exitcode == 0


def test_loader_shards_the_dataset_by_rank(tmp_path):
    """N ranks x B samples per step are N*B DISTINCT samples: rank r's batch i = images [(i*world + r)*B, +B)
    (round-2 advisor finding: every rank used to read the same batch).  Listing and index arithmetic need no GPU."""
    from shmgan_amd.data import PSD_SUBDIRS, PolarDataset
    n, B, world = 13, 2, 3
    for sub in PSD_SUBDIRS:
        (tmp_path / sub).mkdir()
        for i in range(n):
            (tmp_path / sub / f"im_{i:03d}.png").write_bytes(b"")
    seen = []
    for r in range(world):
        ds = PolarDataset(str(tmp_path), 32, batch_size=B, rank=r, world=world)
        assert len(ds) == n // (B * world) == 2
        seen += [ds.image_index(i, b) for i in range(len(ds)) for b in range(B)]
    assert sorted(seen) == list(range(len(seen))) and len(seen) == 12          # disjoint, contiguous, no sample twice
    assert PolarDataset(str(tmp_path), 32, batch_size=B, rank=0, world=1).image_index(3, 1) == 7
