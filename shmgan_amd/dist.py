"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The reference has no distributed code; the path shards naturally because
samples are independent (InstanceNorm, no BatchNorm).  Each rank runs the same step on its own
B samples; the two flat gradient buckets (D: 26.4 MB, G: 74.1 MB at S=256) are summed with
all-reduce on a side stream and scaled by 1/world inside the clip+Adam kernel, which makes
N ranks x B samples equal to one step on N*B samples under the batch rule (mean over samples).
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchange_active():
    """Does train_step exchange gradients?  With more than one rank always; with a ONE-rank process group only under SHM_DP_FORCE=1 --
    the rehearsal a 1-GPU box allows: every bucket goes through RCCL's all-reduce on the side stream, behind the same events, as it
    would on 8 GPUs (the sum over one rank changes nothing), so the nccl branch of this module runs under test before the first
    multi-GPU job does (tests/test_dist_gpu.py::test_one_rank_rccl_exchange...)."""
    w = world_size()
    return w > 1 or (w == 1 and dist.is_available() and dist.is_initialized() and os.environ.get("SHM_DP_FORCE") == "1")


class GradExchangeError(RuntimeError):
    """A gradient collective failed or timed out (a peer rank died or hung).  The step cannot be completed: the process
    must end with a non-zero exit code so that the launcher tears the job down -- never retried, never re-exec'ed."""


def init_process_group(backend="nccl", device=None, timeout_s=None, **kw):
    """torch.distributed.init_process_group with a failure path for the gradient exchange: a finite collective timeout
    (SHM_DP_TIMEOUT_S, default 300 s; the library default for nccl is 10 min, for gloo 30 min) and, for nccl (= RCCL), the
    watchdog's asynchronous error handling, so that a rank whose peer has died aborts with an error instead of
    waiting in all_reduce forever.  Rendezvous defaults to 127.0.0.1 (one node; the container hostname may not resolve)."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("SHM_DP_TIMEOUT_S", "300"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")        # tear the process down on a failed / timed-out collective
        if device is not None:
            kw.setdefault("device_id", torch.device(device))
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)


class GradReducer:
    """Sums flat gradient buckets across ranks.  On GPU tensors the collective is issued on a
    dedicated stream (it waits for the producing stream through an event, and hands back an event
    the consumer waits on), so the D bucket overlaps the generator backward.  On CPU tensors
    (gloo, used by the tests) it is a plain blocking all-reduce."""

    def __init__(self, device=None):
        self.device = device
        self.stream = None
        # measurement only (bench.py, N > 1): when a list, every collective is bracketed by timing events on the reducer stream
        # and every consumer-side wait by timing events on the consumer stream; comm_summary() turns them into milliseconds
        self.probe = None

    def allreduce_async(self, flat, after=None, tag="g"):
        """`after`: optional extra event (e.g. the wgrad lane) the collective must also wait for.  tag: bucket family ("d" / "g")
        for the probe."""
        if not exchange_active():
            return None
        if not flat.is_cuda:
            self._reduce(flat)
            return None
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=flat.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        done = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if after is not None:
                self.stream.wait_event(after)
            if self.probe is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.stream)          # behind the waits: the bracket holds the collective, not the producer
                self._reduce(flat)
                t1.record(self.stream)
                self.probe.append((tag, flat.numel() * flat.element_size(), t0, t1))
            else:
                self._reduce(flat)
            done.record(self.stream)
        return done

    def wait_on(self, event, stream=None):
        """Consumer side: `stream` (default: the current one) waits for a collective's `done` event.  Under the probe the wait is
        bracketed by two timing events on that stream: their distance is the time the consumer stream sat idle behind the
        collective, i.e. the EXPOSED communication time of this wait."""
        if event is None:
            return
        stream = stream or torch.cuda.current_stream()
        if self.probe is None:
            stream.wait_event(event)
            return
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        stream.wait_event(event)
        t1.record(stream)
        self.probe.append(("wait", 0, t0, t1))

    def comm_summary(self, steps):
        """Per-step averages of the probe (call after a device synchronize): collective time on the reducer stream per bucket
        family, bytes reduced, and the exposed time (consumer-stream waits).  Clears the probe."""
        out = {"exposed_ms": 0.0, "d_bucket_ms": 0.0, "g_buckets_ms": 0.0, "bytes": 0, "collectives": 0}
        for tag, nbytes, t0, t1 in self.probe or ():
            ms = t0.elapsed_time(t1)
            if tag == "wait":
                out["exposed_ms"] += ms
            else:
                out["d_bucket_ms" if tag == "d" else "g_buckets_ms"] += ms
                out["bytes"] += nbytes
                out["collectives"] += 1
        n = max(int(steps), 1)
        out = {k: (round(v / n, 4) if isinstance(v, float) else v // n) for k, v in out.items()}
        if self.probe is not None:
            self.probe = []
        return out

    @staticmethod
    def _reduce(flat):
        """all-reduce(sum); any backend error (peer gone, timeout, aborted communicator) becomes GradExchangeError."""
        try:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        except Exception as e:          # torch raises RuntimeError / DistBackendError / DistNetworkError depending on the backend
            raise GradExchangeError(f"rank {dist.get_rank()}: all-reduce of a {flat.numel()}-element gradient bucket failed: "
                                    f"{type(e).__name__}: {e}") from e

    @staticmethod
    def wait(event):
        if event is not None:
            torch.cuda.current_stream().wait_event(event)
