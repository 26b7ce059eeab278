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
