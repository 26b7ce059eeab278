"""Drop-in for the reference trainer object on MI355X.

`ShmGANwithSSpecSeg` mirrors the surface of /root/reference/ShmGANwithSSpecSeg.py:86-1309
that main.py / test.py use for the training hot path: `__init__(args)`, `build_generator()`,
`build_discriminator()`, `train_step(orig0, orig45, orig90, orig135, origED)`,
`custom_per_image_standardization`, `gram_matrix`, the `optimizer_G/D` iteration counters and
the loss / image attributes `train_step` leaves behind (SHM.py:467-875).

Semantics are the reference's "as executed" ones (SURVEY.md findings 3-7, oracle/step_torch.py
restates them): batch rule for B>1 = every sample is an independent B=1 reference step and
the loss is the mean over samples; the style-loss factor is explicit (`style_factor=`); RNG
draws default to fresh ones but can be injected (`draws=`) for parity tests.

One process drives one GPU.  Under torch.distributed (backend nccl = RCCL) every rank holds
a replica; gradients are summed with all-reduce on a side stream (D bucket overlapped with
the generator backward) and scaled by 1/world inside the clip+Adam kernel.
"""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .dist import GradReducer, exchange_active, world_size
from .model import Arena, Discriminator, Generator, WgradLane, pad_channels
from .specseg import SpecSeg


class _Optimizer:
    """Keras adam_v2.Adam(learning_rate=ExponentialDecay(lr0, 10000, 0.95)) bookkeeping
    (SHM.py:169-175); the update itself is shm_adam_clip."""

    def __init__(self, lr0, beta_1, beta_2, epsilon=1e-7):
        self.lr0, self.beta_1, self.beta_2, self.epsilon = lr0, beta_1, beta_2, epsilon

    def alpha(self, iterations):
        t = iterations + 1
        lr = self.lr0 * 0.95 ** (iterations / 10000.0)
        return lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply(self, P, gscale=1.0):
        ops.adam_clip(P.flat, P.m, P.v, P.grad, P.n, self.alpha(P.iterations), self.beta_1, self.beta_2,
                      self.epsilon, gscale)
        P.iterations += 1


_DEFAULTS = dict(image_size=128, batch_size=1, filter_size=64, g_lr=0.00002, d_lr=0.00002, beta1=0.5, beta2=0.99,
                 c_dim=5, num_epochs=200, num_iteration_decay=100000, n_critic=5, d_repeat_num=6, mode="train",
                 data_dir="", model_save_dir="./models", checkpoint_save_dir="./checkpoints", result_dir="./results",
                 log_dir="./logs/train", log_step=1, checkpoint_save_step=10)

LOSS_NAMES = ["total_Generator_loss", "total_Discriminator_loss", "total_Classification_loss", "G_gan_loss",
              "G_clsf_loss", "D1_RealFake_loss", "D3_RealFake_cyc", "D2_RealFake_target", "D4_RealFake_cyc",
              "D1_classification_loss", "D3_classification_loss", "D4_classification_loss", "L1_loss_Gen",
              "ssim_cyc_loss", "content_loss", "style_loss", "total_NST_loss", "Spec_loss"]


class KernelAbortError(RuntimeError):
    """A kernel of an earlier step gave up (a barrier of the one-pass InstanceNorm backward timed out) and went on with wrong numbers.
    The optimizer kernels refused every update from that moment on (shm_set_abort_words), so the weights are those of the last good
    step; the run must not continue."""


_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "f32": torch.float32,
           "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


class ShmGANwithSSpecSeg:
    def __init__(self, args=None, device=None, compute_dtype="float32", grad_dtype=None, attention="executed",
                 xent_mode="executed", **overrides):
        """compute_dtype: "float32" (the reference's precision: exact-fp32 MFMA) or "bfloat16" (BASELINE
        configs 4-5: activations and MFMA operands in bf16, fp32 accumulation, fp32 master weights,
        statistics, losses, weight gradients and Adam).
        attention: "executed" (default) = the graph the reference runs: attention_layer is evaluated once on a constant zero
        mask at build time, so the skips get + 0 (SURVEY finding 3); "live" = the architecture as intended: every step's
        SpecSeg mask goes through attention_layer (MaxPool -> 2 x Conv3x3 + LeakyReLU, SHM.py:404-412) into the four
        generator skips (SHM.py:290-293) and the discriminator (SHM.py:359), its 10 convolutions are trained."""
        assert attention in ("executed", "live")
        self.attention = attention
        # "executed" (default): the classification-loss gradient as TensorFlow computes it -- the fused
        # softmax_cross_entropy_with_logits kernel hands back softmax - labels, which for D1's un-normalised label row
        # [0,0,0,0,TARGET_LABELS] (SHM.py:477, 688, 702) is not the derivative (SURVEY-style finding 8);
        # "intended": the true derivative TARGET_LABELS * (softmax - onehot).  Loss values do not depend on the mode.
        assert xent_mode in ("executed", "intended")
        self.xent_mode = xent_mode
        self.compute_dtype = _DTYPES[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        # gradient-signal tensors between an input-gradient product and the next IN/LeakyReLU backward:
        # default = compute_dtype; "float32" with bf16 compute selects SHM_BF16_GF32 (include/shmgan_hip.h)
        self.grad_dtype = _DTYPES[grad_dtype] if isinstance(grad_dtype, str) else (grad_dtype or self.compute_dtype)
        if self.compute_dtype == torch.float32:
            self.grad_dtype = torch.float32
        self.pad = pad_channels(self.compute_dtype)
        a = dict(_DEFAULTS)
        if args is not None:
            a.update({k: v for k, v in vars(args).items()})
        a.update(overrides)
        self.args = SimpleNamespace(**a)
        for k in ("c_dim", "image_size", "batch_size", "num_epochs", "num_iteration_decay", "g_lr", "d_lr", "n_critic",
                  "beta1", "beta2", "d_repeat_num", "mode", "data_dir", "model_save_dir", "checkpoint_save_dir",
                  "result_dir", "log_dir", "log_step", "checkpoint_save_step", "filter_size"):
            setattr(self, k, a[k])
        # SHM.py:157-212
        self.seed, self.randomness, self.dropout_amnt = 25, 0.50, 0.2
        self.TARGET_LABELS = 0.90
        self.c_dim = 5
        self.epoch, self.train_G_after = 0, 0
        self.stddev_arr, self.mean_arr, self.variance_arr = [], [], []
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("shmgan_amd needs an MI355X (ROCm) device: there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        # both optimisers use the schedule built from g_lr (SHM.py:169-174; d_lr is unused there)
        self.optimizer_G = _Optimizer(self.g_lr, self.beta1, self.beta2)
        self.optimizer_D = _Optimizer(self.g_lr, self.beta1, self.beta2)
        self.arena = Arena(self.device)
        self._ws = None
        self._lane = None
        self.G = self.D = self.SpecSeg = None
        # zeros at graph-build time (SHM.py:206; the attention convs only ever see this constant, finding 3);
        # train_step overwrites it with SpecSeg.predict(I90_Ych) (SHM.py:492), which feeds Spec_loss only
        self.specular_candidate = None
        self._rng = np.random.default_rng(self.seed)
        self._draw_count = 0
        self._reducer = GradReducer(self.device)
        self._prefetched = None          # _prologue() of the next batch, issued by train_step(next_batch=)
        self._loss_cache = None
        # abort words (include/shmgan_hip.h, shm_set_abort_words): a device word the optimizer kernel checks before it touches the
        # weights, and a word in pinned host memory the host polls without synchronising (_check_abort)
        self._abort_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._abort_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._abort_np = self._abort_host.numpy()
        # test diagnostics: called with no arguments between the last forward pass and the first backward kernel of a step
        # (tests/test_step_gpu.py pins LeakyReLU signs there; never set on the hot path)
        self.before_backward = None
        # SHM_FORWARD_PARTS=2: the two big forward passes (cyclic generator pass on 5B images, discriminator on 12B) as two half
        # batches on the two streams (samples are independent).  Measured in round 3 and NOT the default: in lockstep the halves
        # gain 0.2 ms of a 121 ms fp32 step (within noise), taking turns convolution by convolution (so that one half's normalisation
        # pass runs under the other's convolution) LOSES 2 ms -- a single stream already fills the launch tails
        import os
        self.forward_parts = int(os.environ.get("SHM_FORWARD_PARTS", "1"))
        # the D-loss backward on the second stream (train_step): measured on one box, A/B by the variable, bf16 22.48 -> 22.25 ms, S = 512 B = 4
        # 42.79 -> 42.58, B = 32 79.3 -> 78.6; fp32 114.2 -> 114.2 (MFMA bound either way) -- default on in bf16, off in fp32
        self.d_fwd_on_lane = os.environ.get("SHM_D_FWD_LANE", "1" if self.compute_dtype != torch.float32 else "0") == "1"
        self.img_loss_on_lane = os.environ.get("SHM_IMG_LOSS_LANE", "1" if self.compute_dtype != torch.float32 else "0") == "1"
        self.d_bwd_on_lane = os.environ.get("SHM_D_BWD_LANE", "1" if self.compute_dtype != torch.float32 else "0") == "1"
        self.style_factor = 1.0 / float(2 * 9 * self.image_size * self.image_size) ** 2   # as intended (finding 7)

    # ------------------------------------------------------------------ workspace
    def _workspace(self, nbytes):
        """Split-K slab workspace of the wgrad launches (they run in order on one stream)."""
        n = (nbytes + 3) // 4
        if self._ws is None or self._ws.numel() < n:
            # 128 MiB covers every layer at S=256/B=8; growing mid-step would hand memory still in
            # use on the wgrad lane back to the allocator, hence the join before re-allocating
            if self._lane is not None:
                self._lane.join()
                torch.cuda.current_stream().synchronize()
            self._ws = torch.empty(max(n, 32 << 20), dtype=torch.float32, device=self.device)
        return self._ws

    def _arm_abort(self):
        """Point the library's abort words at this trainer's pair (per-thread state of the library: several trainers may live in
        one process, the one that steps owns them)."""
        ops.set_abort_words(self._abort_dev, self._abort_host)

    def _check_abort(self, sync=False):
        """Raise KernelAbortError if a kernel of this trainer gave up.  Without `sync` this is one read of host memory (the kernel
        wrote the pinned word itself), so it costs the step nothing; it sees every abort whose kernel has finished -- the host runs
        ahead of the device, so an abort of step t may only be seen at the top of step t + 2, and the device-side check in
        shm_adam_clip is what keeps the weights clean in between.  sync=True (losses(), checkpoints, release()) waits for the device
        first and is exact."""
        if self._abort_host is None:
            return
        if sync:
            torch.cuda.synchronize(self.device)
        if int(self._abort_np[0]) != 0 or (sync and int(self._abort_dev.item()) != 0):
            names = self.arena.fused_timeouts()
            raise KernelAbortError(
                f"in_bwd_fused8_kernel: a group barrier timed out (scratch {names or 'unknown'}): the step's gradients are built on "
                "unfinished sums.  No optimizer update has been applied since (shm_adam_clip checks the abort word on the device); "
                "the weights are those of the last good step.  Set SHM_ELEM_FUSED_BWD=0 (two-pass InstanceNorm backward) and report; "
                "clear_abort() re-arms this trainer")

    def clear_abort(self):
        """Clear the abort words (after a KernelAbortError, once its cause is dealt with)."""
        torch.cuda.synchronize(self.device)
        self._abort_dev.zero_()
        self._abort_np[0] = 0
        self.arena.fused_timeouts()
        torch.cuda.synchronize(self.device)

    def release(self):
        """Drop every device buffer this trainer holds (arena, split-K workspace, both models and their optimizer state) so that
        another trainer can be built in the same process (bench.py's extra configurations); the object is unusable afterwards.
        Raises KernelAbortError (after releasing) if a kernel of this trainer gave up and nobody has heard of it yet."""
        if self._lane is not None:
            self._lane.join()
        torch.cuda.synchronize(self.device)
        aborted = self._abort_host is not None and (int(self._abort_np[0]) != 0 or int(self._abort_dev.item()) != 0)
        ops.set_abort_words(None, None)
        self._abort_dev = self._abort_host = self._abort_np = None
        self.arena.t.clear()
        self._ws = self._prefetched = self._loss_cache = None
        self.G = self.D = self.SpecSeg = None
        self.specular_candidate = None
        if aborted:
            raise KernelAbortError("in_bwd_fused8_kernel: a group barrier timed out during this trainer's last steps (seen at release())")

    def _get_lane(self):
        import os
        if self._lane is None:
            self._lane = WgradLane(self.device, enabled=os.environ.get("SHM_NO_WGRAD_LANE") is None)
        return self._lane

    # ------------------------------------------------------------------ builders
    def build_generator(self):
        """SHM.py:228-327."""
        return Generator(self.image_size, self.filter_size, self.device, self.arena, self._workspace, self._get_lane(),
                         dtype=self.compute_dtype, grad_dtype=self.grad_dtype, attention=self.attention == "live")

    def build_discriminator(self):
        """SHM.py:343-380."""
        return Discriminator(self.image_size, self.filter_size, self.device, self.arena, self._workspace,
                             self.dropout_amnt, self._get_lane(), dtype=self.compute_dtype, grad_dtype=self.grad_dtype,
                             attention=self.attention == "live")

    def build_specseg(self):
        """SHM.py:930-931: SpecSeg(image_size, image_size, 1) then load_model('specsegv3_chkpt.h5').  The
        checkpoint is not available, so the network starts from the initialisers of SpecSeg.py; load trained
        Keras weights with `self.SpecSeg.set_weights(keras_model.get_weights())`."""
        return SpecSeg(self.image_size, self.device, self.arena).init_random()

    def build(self, seed=42, beta_seed=43, attention_seed=45):
        """Build G, D and SpecSeg and give G/D the synthetic init of SURVEY 8(d): weights N(0,0.02) from
        default_rng(seed) (RandomNormal(0,0.02), SHM.py:200), biases 0, IN beta N(0,0.02); the live attention branch's
        kernels N(0,0.02) from default_rng(attention_seed), generator levels first, then the discriminator's."""
        self.G = self.build_generator()
        self.D = self.build_discriminator()
        if self.SpecSeg is None:
            self.SpecSeg = self.build_specseg()
        rng = np.random.default_rng(seed)
        ng, nd = 2 * len(self.G.layers), 7
        gw = [np.zeros(s, np.float32) if len(s) == 1 else rng.normal(0.0, 0.02, s).astype(np.float32)
              for s in self.G.P.shapes[:ng]]
        dw = [rng.normal(0.0, 0.02, s).astype(np.float32) for s in self.D.P.shapes[:nd]]
        if self.attention == "live":
            arng = np.random.default_rng(attention_seed)
            ga = [(arng.normal(0.0, 0.02, s) if len(s) == 4 else np.zeros(s)).astype(np.float32) for s in self.G.P.shapes[ng:]]
            da = [(arng.normal(0.0, 0.02, s) if len(s) == 4 else np.zeros(s)).astype(np.float32) for s in self.D.P.shapes[nd:]]
            self.G.P.load(gw + ga)
            self.D.P.load(dw + da)
            self.G.weights_dirty = self.D.weights_dirty = True
            gw = dw = None
        brng = np.random.default_rng(beta_seed)
        gb = [brng.normal(0.0, 0.02, (c,)).astype(np.float32) for c in self.G.in_channels]
        db = [brng.normal(0.0, 0.02, (c,)).astype(np.float32) for c in self.D.chan[1:]]
        if gw is not None:
            self.G.set_weights(gw)
            self.D.set_weights(dw)
        self.G.set_betas(gb)
        self.D.set_betas(db)
        return self

    # ------------------------------------------------------------------ small reference helpers
    def custom_per_image_standardization(self, image):
        """SHM.py:1271-1309 on an already-YUV [B,S,S,3] tensor: x / max(std, 1/256), statistics
        per sample, no mean subtraction.  API-parity helper for test.py:218 (off the hot path, plain
        tensor arithmetic); train_step uses the fused rgb->yuv+standardise kernels (`preprocess`)."""
        x = self._dev(image)
        m = x.mean(dim=(1, 2, 3))
        var = torch.relu((x * x).mean(dim=(1, 2, 3)) - m * m)
        scale = torch.clamp(torch.sqrt(var), min=1.0 / 256.0)
        self.stddev_arr = list(scale)
        return x / scale.view(-1, 1, 1, 1)

    def preprocess(self, rgb, name):
        """tf.image.rgb_to_yuv + custom_per_image_standardization (SHM.py:480-484, 1271-1309)."""
        B, S = rgb.shape[0], self.image_size
        yuv = self.arena.get(f"pre/yuv/{name}", (B, S, S, 3))
        acc = self.arena.get(f"pre/acc/{name}", (B * 2,), torch.float64)
        scale = self.arena.get(f"pre/scale/{name}", (B,))
        ops.rgb2yuv_std(rgb, yuv, acc, scale, B, S * S)
        return yuv, scale

    def _prologue(self, orig, slot):
        """The weight-independent head of a step (SHM.py:480-505 + the SpecSeg.predict of SHM.py:492): rgb->yuv +
        standardisation of the five views, the CbCr average and the specular mask (on the second stream).  Nothing here
        reads G's or D's weights, so train_step(next_batch=) issues it for step t+1 while step t's last gradient bucket is
        still being all-reduced ("the next batch's weight-independent prologue"); buffers are double-buffered
        by `slot`."""
        A, S = self.arena, self.image_size
        B = orig[0].shape[0]
        ds, scales = [], []
        for k in range(5):
            y, sc = self.preprocess(orig[k], f"{k}/s{slot}")
            ds.append(y)
            scales.append(sc)
        cbcr = A.get(f"pre/cbcr/s{slot}", (B, S, S, 2))
        ops.avg_cbcr(ds, cbcr, B * S * S)
        # specular mask: nothing on the gradient path reads it, so it runs on the second stream beside the generator forward
        lane = self._get_lane()
        if self.SpecSeg is None:
            self.SpecSeg = self.build_specseg()
        box = {}
        lane.submit(lambda: box.__setitem__("mask", self.SpecSeg.forward_plane(ds[2], 3, 0, B, tag=f"specseg/step{slot}")))
        return SimpleNamespace(slot=slot, key=tuple((t.data_ptr(), tuple(t.shape)) for t in orig), orig=orig, ds=ds,
                               scales=scales, cbcr=cbcr, mask=box["mask"])

    def gram_matrix(self, x):
        """SHM.py:1176-1180 (host-side convenience on a torch tensor; not on the hot path)."""
        return torch.einsum('bijc,bijd->bcd', x, x) / float(x.shape[1] * x.shape[2])

    # ------------------------------------------------------------------ draws
    def _default_draws(self, B):
        """Fresh draws of a step: the five RNG flags (SHM.py:509-513) on the host; GaussianNoise(0.1) for the D1 / D2 inputs
        (SHM.py:352) and the Dropout keep mask (SHM.py:363) on the device (shm_randn / shm_keep_mask: Philox keyed by the
        trainer's seed, the step counter and the rank, so replicas draw different noise and agree on the flags)."""
        S, s = self.image_size, self.image_size // 32
        flags = tuple(bool(u < self.randomness) for u in self._rng.random(5))
        noise = self.arena.get("draws/noise", (2 * B, S, S, 3))
        keep = self.arena.get("draws/keep", (2 * B, s, s, 16 * self.filter_size))
        self._draw_count += 1
        seed = (self.seed << 32) | (self._draw_count & 0xFFFFFFFF)
        ops.randn(noise, 0.1, seed, 2 * self._rank())
        ops.keep_mask(keep, self.dropout_amnt, seed, 2 * self._rank() + 1)
        return SimpleNamespace(flags=flags, target_label=float(self.TARGET_LABELS), noise=noise, keep_mask=keep)

    @staticmethod
    def _rank():
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def _dev(self, t):
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return t.to(self.device, torch.float32).contiguous()

    # ------------------------------------------------------------------ data parallel
    def _world(self):
        return world_size()

    def _allreduce_async(self, flat, after=None, tag="g"):
        """Sum `flat` over ranks on the side stream; returns an event to wait on (or None)."""
        return self._reducer.allreduce_async(flat, after, tag)

    # ------------------------------------------------------------------ the step
    def train_step(self, orig0, orig45, orig90, orig135, origED, *, draws=None, style_factor=None, apply=True, next_batch=None):
        """One SHM.py:467-875 step: D update + G update from a 5-view batch [B,S,S,3] in [0,1].
        next_batch: the five tensors of the FOLLOWING step (or a callable returning them, called late in this step's issue
        order; None when it returns None): their weight-independent prologue (_prologue) is issued before this step waits
        for its last gradient collective, and the next train_step on those same tensors (same storage, not modified in
        between) picks the result up instead of recomputing it."""
        if self.G is None:
            self.build()
        self._arm_abort()
        self._check_abort()              # an earlier step's kernel gave up: stop here, before anything else is issued
        G, D, A = self.G, self.D, self.arena
        S, F = self.image_size, self.filter_size
        orig = [self._dev(t) for t in (orig0, orig45, orig90, orig135, origED)]
        B = orig[0].shape[0]
        npix = S * S
        if draws is None:
            draws = self._default_draws(B)
        flags = [bool(f) for f in draws.flags]
        fmask = sum(1 << k for k in range(5) if flags[k])
        T = float(draws.target_label)
        noise = self._dev(draws.noise)
        keep = self._dev(draws.keep_mask)
        sf = float(self.style_factor if style_factor is None else style_factor)
        world = self._world()

        G.zero_grad()
        D.zero_grad()
        G.prepare_weights()
        D.prepare_weights()

        # ---- pre-processing (outside the tape, SHM.py:480-505) and the specular mask SpecSeg.predict(I90_Ych) (SHM.py:492):
        # taken from the previous step's look-ahead when it was given these tensors, issued here otherwise
        lane = self._get_lane()
        pro, self._prefetched = self._prefetched, None
        if pro is None or pro.key != tuple((t.data_ptr(), tuple(t.shape)) for t in orig):
            pro = self._prologue(orig, 0 if pro is None else pro.slot)
        ds, scales, cbcr = pro.ds, pro.scales, pro.cbcr
        self.specular_candidate = pro.mask

        # ---- live attention branch: the maps of this step's mask, shared by the six G calls and the twelve D calls
        attn_g = attn_d = None
        if self.attention == "live":
            lane.join()                                    # the mask comes from the second stream
            attn_g = G.attention_forward(self.specular_candidate, B)
            attn_d = D.attention_forward(self.specular_candidate, B)

        adt, PAD_C = self.compute_dtype, self.pad
        # D batch layout: [D1: B][D3: 5B][D2: B][D4: 5B]
        xd = A.get_slack("d/x16", (12 * B, S, S, D.in_pitch), adt, D.pad)          # one 16-byte chunk per pixel (Discriminator.in_pitch) + a K-row of slack
        # The REAL half of the discriminator batch (D2, D4: SHM.py:563, 638-642) depends on no generator output: in bf16 its five convolution
        # blocks run on the second stream beside the generator's forward passes (round 6); the heads wait for `ev_dreal`
        ev_dreal = None
        d_real_early = self.d_fwd_on_lane and lane.stream is not None and not (self.forward_parts > 1)
        if d_real_early:
            ops.pack_rgb16(orig[4], noise[B:2 * B], xd[6 * B:7 * B], B * npix)                 # D(2) SHM.py:563
            for k in range(5):                                                                 # D(4) SHM.py:638-642
                ops.pack_rgb16(orig[k], None, xd[(7 + k) * B:(8 + k) * B], B * npix)
            D.start(xd, keep, [(0, B, 0), (6 * B, B, B)], attn_d)
            lane.submit(lambda: D.trunk_rows(6 * B, 12 * B))
            ev_dreal = lane.event()

        # ---- G(1)  SHM.py:517-538
        gen_in = A.get("g1/in", (B, S, S, PAD_C), adt)
        ops.build_gen_input(ds, None, fmask, 0, gen_in, B, npix)
        gen_Y = G.forward(gen_in, "g1", attn=attn_g)

        gen_rgb = A.get("g1/rgb", (B, S, S, 3))
        ops.yuv2rgb(gen_Y, cbcr, noise[:B], gen_rgb, xd[0:B], B, B, npix)                  # SHM.py:544-559

        # ---- G(2): cyclic  SHM.py:576-624
        cyc_in = A.get("cyc/in", (5 * B, S, S, PAD_C), adt)
        ops.build_gen_input(ds, gen_Y, fmask, 1, cyc_in, B, npix)
        cyc_Y = G.forward(cyc_in, "cyc", attn=attn_g, parts=self.forward_parts)
        cyc_rgb = A.get("cyc/rgb", (5 * B, S, S, 3))
        ops.yuv2rgb(cyc_Y, cbcr, None, cyc_rgb, xd[B:6 * B], 5 * B, B, npix)
        if not d_real_early:
            ops.pack_rgb16(orig[4], noise[B:2 * B], xd[6 * B:7 * B], B * npix)                 # D(2) SHM.py:563
            for k in range(5):                                                                 # D(4) SHM.py:638-642
                ops.pack_rgb16(orig[k], None, xd[(7 + k) * B:(8 + k) * B], B * npix)

        # ---- image-space losses (L1, SSIM, NST: SHM.py:744-826) and their gradients wrt the two Y planes.  Nothing reads them before the
        # G-loss backward leaves the discriminator (conv3x3_dgrad_sum1 below accumulates into dgen_y / dcyc_y), so in bf16 they run on the second
        # stream beside the discriminator's forward pass and the G-loss backward through it (round 6: ~0.5 / 1.3 / 1.4 ms of SSIM and L1 passes
        # were exposed per step at B = 8 / S = 512 B = 4 / B = 32); the main stream waits for `ev_img` where it needs them
        il = A.get("loss/img", (32,), torch.float64)
        dgen_y = A.get("loss/dgen_y", (B, S, S, 1))
        dcyc_y = A.get("loss/dcyc_y", (5 * B, S, S, 1))
        ws = self._img_ws(B)
        optr = (C.c_void_p * 5)(*[t.data_ptr() for t in orig])
        dptr = (C.c_void_p * 5)(*[t.data_ptr() for t in ds])
        ev_img = None
        if self.img_loss_on_lane and lane.stream is not None:
            lane.submit(lambda: ops.image_losses(gen_rgb, cyc_rgb, cyc_Y, cbcr, optr, dptr, fmask, sf, il, dgen_y, dcyc_y, ws, B, S))
            ev_img = lane.event()
        else:
            ops.image_losses(gen_rgb, cyc_rgb, cyc_Y, cbcr, optr, dptr, fmask, sf, il, dgen_y, dcyc_y, ws, B, S)

        # ---- D on all 12B images (noise + dropout on the D1 and D2 slices only)
        dparts = None
        if self.forward_parts > 1 and lane.stream is not None:     # fake half on the main stream, real half on the second one
            dparts = [(6 * B, 12 * B, lane.submit), (0, 6 * B, lambda fn: fn())]
        if d_real_early:
            D.trunk_rows(0, 6 * B)                                  # the fake half, here; the real half is already under way on the second stream
            torch.cuda.current_stream().wait_event(ev_dreal)
            rf, cls = D.heads()
        else:
            rf, cls = D.forward(xd, keep, [(0, B, 0), (6 * B, B, B)], parts=dparts, attn=attn_d, join=lane.join if dparts else None)
        np_ = (S // 32) ** 2

        # ---- losses  SHM.py:669-844
        dl = A.get("loss/dhead", (16,), torch.float64)
        drf_d = A.get("loss/drf_d", (12 * B, np_))
        dcls_d = A.get("loss/dcls_d", (12 * B, 5))
        drf_g = A.get("loss/drf_g", (6 * B, np_))
        ops.dhead_losses(rf, cls, dl, drf_d, dcls_d, drf_g, B, np_, T,
                         ops.XENT_TF_FUSED if self.xent_mode == "executed" else ops.XENT_INTENDED)
        sl = A.get("loss/spec", (5,), torch.float64)                      # Spec_loss, logged only  SHM.py:792-806
        lane.submit(lambda: ops.spec_loss(cyc_Y, cbcr, dptr, self.specular_candidate, sl, B, npix))

        # ---- D backward (weights) then its all-reduce overlapped with everything below
        if self.before_backward is not None:
            self.before_backward()
        # The D-loss backward (weights only) depends on nothing below and nothing below depends on it: the whole chain -- InstanceNorm backward,
        # stride-2 input gradients, weight gradients -- goes to the second stream, in front of the generator's weight gradients, and the main
        # stream goes straight on to the G-loss backward (round 6; SHM_D_BWD_LANE=0 keeps it on the main stream).  Its buffers are keyed by the
        # batch (12 B here, 6 B for the data gradient below), the split-K workspace is used in lane order.
        if self.d_bwd_on_lane and lane.stream is not None:
            lane.submit(lambda: D.backward_params(drf_d, dcls_d))
        else:
            D.backward_params(drf_d, dcls_d)
        ev_d = self._allreduce_async(D.P.grad, after=self._get_lane().event(), tag="d")

        # ---- G-loss gradient through D (data gradient only), then G backward
        # Both first layers are left through the channel-summed stencil (ops.conv3x3_dgrad_sum1): yuv_to_rgb's
        # backward needs r+g+b of the image gradient, the G o G chain the sum over the substituted view channels.
        FD = D.chan[1]
        dzd = D.backward_input_dz(6 * B, drf_g)
        weff_d = A.get("d/weff", (9, FD))
        ops.sum_input_channels(D.P.vars[0], 3, FD, 0b111, weff_d)
        if ev_img is not None:
            torch.cuda.current_stream().wait_event(ev_img)          # dgen_y / dcyc_y of the image losses
        ops.conv3x3_dgrad_sum1(dzd[0:B], FD, weff_d, dgen_y, 1, B, S, S, FD, 2, 1)
        ops.conv3x3_dgrad_sum1(dzd[B:6 * B], FD, weff_d, dcyc_y, 1, 5 * B, S, S, FD, 2, 1)
        dzg = G.backward(dcyc_y, "cyc", need_dx="dz")
        weff_g = A.get("g/weff", (5, 9, F))
        for k in range(5):                                       # G o G chain  SHM.py:576-580
            ops.sum_input_channels(G.P.vars[0], 10, F, sum(1 << j for j in range(5) if j != k and flags[j]), weff_g[k])
        ops.conv3x3_dgrad_sum1(dzg, F, weff_g, dgen_y, 5, B, S, S, F, 1, 1)
        update_g = self.epoch >= self.train_G_after
        # no collective for a gradient nobody applies (it would also still be in flight when the next step
        # zeroes the bucket)
        reduce_g = exchange_active() and (update_g or not apply)
        # The G(1) backward is the last pass that touches the generator's weight gradients, and it finishes the layers last
        # to first: each stage's slice of the flat gradient is all-reduced as soon as the weight gradient of its lowest
        # layer has been issued on the wgrad lane (the collective waits for that lane event on the reducer stream), under
        # the rest of the pass.  Only the small last bucket (first encoder layers + head kernel + biases) is exposed.
        ev_g = None
        on_wgrad = None
        if reduce_g:
            plan = {trig: slices for trig, slices in G.grad_buckets()}

            def on_wgrad(li):
                for lo, hi in plan.get(li, ()):
                    self._allreduce_async(G.P.grad[lo:hi], after=lane.event())
        G.backward(dgen_y, "g1", need_dx=False, on_wgrad=on_wgrad)
        if attn_g is not None:
            G.attention_backward()                      # the skip gradients of both passes, summed per sample
        G.finish_grads()
        lane.join()                             # all weight gradients (both models) are complete
        if reduce_g:
            for lo, hi in plan[None]:
                ev_g = self._allreduce_async(G.P.grad[lo:hi])        # the reducer stream runs its collectives in order

        # ---- look-ahead: the next batch's weight-independent prologue goes in front of the waits on this step's collectives
        if next_batch is not None:
            nb = next_batch() if callable(next_batch) else next_batch
            if nb is not None:
                self._prefetched = self._prologue([self._dev(t) for t in nb], pro.slot ^ 1)

        # ---- clip + Adam  SHM.py:859-872
        if apply:
            self._reducer.wait_on(ev_d)
            self.optimizer_D.apply(D.P, 1.0 / world)
            D.weights_dirty = True
            if update_g:
                self._reducer.wait_on(ev_g)
                self.optimizer_G.apply(G.P, 1.0 / world)
                G.weights_dirty = True
        elif exchange_active():
            for ev in (ev_d, ev_g):
                self._reducer.wait_on(ev)

        # ---- attribute side effects (SHM.py:538-553, 620-624, 863, 872)
        self.gen_input, self.gen_Y, self.gen_rgb = gen_in, gen_Y, gen_rgb
        self.target_img = orig[4]
        (self.cyc_gen0_rgb, self.cyc_gen45_rgb, self.cyc_gen90_rgb, self.cyc_gen135_rgb,
         self.cyc_genED_rgb) = [cyc_rgb[k * B:(k + 1) * B] for k in range(5)]
        self.RealFake_gen_D1, self.label_gen_D1 = rf[0:B], cls[0:B]
        self.RealFake_target_D2, self.label_target_D2 = rf[6 * B:7 * B], cls[6 * B:7 * B]
        self.gradmapD, self.gradmapG = D.P.grads, G.P.grads
        self.stddev_arr = scales          # reference appends forever (a leak); we keep the last step's
        self._last = SimpleNamespace(dl=dl, il=il, sl=sl, npix=npix, B=B, flags=flags, T=T, scales=scales, ds=ds, cbcr=cbcr)
        self._loss_cache = None
        return None

    # ------------------------------------------------------------------ the training loop (SHM.py:889-1139)
    def train(self, args=None, *, max_steps=None, print_fn=print):
        """`shmgan.train(args)` of /root/reference/main.py:107 (loop: SHM.py:889-1139): datasetLoad ->
        build_generator / build_discriminator (+ summaries) -> SpecSeg -> restore the latest checkpoint ->
        for epoch / for batch in range(batches_per_epoch - 1): TARGET_LABELS ~ U(0.8, 1.2) (SHM.py:986),
        next 5-tuple (SHM.py:990), train_step (SHM.py:998) -> checkpoint every `checkpoint_save_step` epochs
        (SHM.py:1125-1128) and once more at the end (SHM.py:1133).

        Differences, all outside the arithmetic: checkpoints are the neutral .npz of save_npz (Keras variable order
        and layouts; newest 3 kept, as CheckpointManager(max_to_keep=3)) instead of TF checkpoints; the summaries go
        to `log_dir`; the Comet histogram upload at step 100 (which raises AttributeError in the reference, SURVEY
        section 3.1) and the per-step gc.collect() are not reproduced.  `max_steps` (tests) stops early.  Returns the
        number of train_step calls made."""
        import os
        import time
        from .data import datasetLoad
        if args is not None:
            for k, v in vars(args).items():
                setattr(self.args, k, v)
                if k in self.__dict__:
                    setattr(self, k, v)
        start = time.perf_counter()
        self.length_dataset, dataset = datasetLoad(self)                       # SHM.py:904
        if self.G is None:
            self.build()                                                       # SHM.py:911-912, 930-931
        # data parallel: every rank trains on its own shard of the dataset (PolarDataset shards by rank) and holds the same
        # weights; files (summaries, checkpoints, log lines) are rank 0's business, the others wait at a barrier
        rank0 = self._rank() == 0
        if not rank0:
            print_fn = lambda *a, **k: None
        if rank0:
            os.makedirs(self.log_dir, exist_ok=True)
            for mdl, fn in ((self.G, "Generator_summary.txt"), (self.D, "Discriminator_summary.txt"), (self.SpecSeg, "SpecSeg_summary.txt")):
                with open(os.path.join(self.log_dir, fn), "w") as f:           # SHM.py:914-919, 933-935
                    mdl.summary(print_fn=lambda x: f.write(x + "\n"))
            os.makedirs(self.checkpoint_save_dir, exist_ok=True)
        self._barrier()
        latest = self._restore_latest()                                        # SHM.py:949-951 (delete_old_checkpoints is False)
        if latest is not None:
            print_fn(f"Latest checkpoint restored!! ({latest})")
        iterator = iter(dataset)                                               # SHM.py:955
        batches_per_epoch = int(self.length_dataset / self.batch_size)         # SHM.py:957
        self.batch_step = 0
        done = False
        total_steps = self.num_epochs * max(batches_per_epoch - 1, 0)
        if max_steps is not None:
            total_steps = min(total_steps, max_steps)
        ahead = []                       # the batch train_step's look-ahead already took from the iterator

        def fetch_next():
            ahead.append(next(iterator, None))
            return ahead[-1]

        for epoch in range(self.num_epochs):                                   # SHM.py:969
            self.epoch = epoch
            print_fn(f"\nStart of Training Epoch {self.epoch}")
            for batch in range(batches_per_epoch - 1):                         # SHM.py:979
                self.batch_step += 1
                self.random_flip = bool(self._rng.random() >= 0.5)            # SHM.py:983 (inert: the map lambda is already traced)
                self.TARGET_LABELS = float(self._rng.uniform(0.8, 1.2))       # SHM.py:986
                element = ahead.pop() if ahead else next(iterator)             # SHM.py:990
                # SHM.py:998; the next tuple is taken from the loader late in this step (its prologue overlaps this step's
                # last gradient collective), not before it: this step must not wait for the next batch's decode
                self.train_step(*element, next_batch=fetch_next if self.batch_step < total_steps else None)
                if max_steps is not None and self.batch_step >= max_steps:
                    done = True
                    break
            if (self.epoch + 1) % self.log_step == 0:                          # SHM.py:1099-1106
                torch.cuda.current_stream().synchronize()
                print_fn("Time taken for epoch {} is {} min\n".format(self.epoch + 1, (time.perf_counter() - start) / 60))
            if (self.epoch + 1) % self.checkpoint_save_step == 0:              # SHM.py:1125-1128
                print_fn("Saving checkpoint for epoch {} at {}".format(self.epoch + 1, self._save_checkpoint()))
            if done:
                break
        print_fn("Saving checkpoint for epoch {} at {}".format(self.epoch + 1, self._save_checkpoint()))   # SHM.py:1133
        return self.batch_step

    def _checkpoints(self):
        import glob
        import os
        return sorted(glob.glob(os.path.join(self.checkpoint_save_dir, "ckpt-*.npz")),
                      key=lambda p: int(os.path.basename(p)[5:-4]))

    def _latest_checkpoint(self):
        c = self._checkpoints()
        return c[-1] if c else None

    def _barrier(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()

    def _restore_latest(self):
        """Load the newest checkpoint that reads back.  ONE rank decides: rank 0 reads every candidate completely (every array
        decoded, i.e. CRC-checked, names and shapes checked against this trainer) WITHOUT touching any state, newest first,
        skipping only files that are damaged as files (zipfile.BadZipFile / EOFError: a run killed mid-write by an older version,
        a full disk); the chosen path (or None) is broadcast, every rank loads exactly that file, and the ranks then agree on a
        success flag -- a rank that could not load what rank 0 chose ends the job (non-zero exit) instead of training on other
        weights, Adam state and draw streams than its peers.  Anything else (EACCES, EIO, a checkpoint of another model
        configuration or attention mode) is not a damaged file: rank 0 broadcasts the reason and EVERY rank raises it (round-4 advisor
        finding: rank 0 used to raise in front of the broadcast and leave its peers waiting in it)."""
        import zipfile
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.G is None:
            self.build()
        chosen, payload, fatal = None, None, None
        if self._rank() == 0:
            try:
                for path in reversed(self._checkpoints()):
                    try:
                        payload = self._read_npz(path)
                        chosen = path
                        break
                    except (zipfile.BadZipFile, EOFError, ValueError) as e:
                        # ValueError: np.load / json on an intact zip member with a damaged .npy header or state string -- a damaged file like
                        # the other two.  _read_npz's own checks (names, shapes, attention mode) raise KeyError, which is not swallowed here
                        print(f"checkpoint {path} is unreadable ({type(e).__name__}: {e}); trying the previous one")
            except Exception as e:                      # not a damaged file: every rank must hear about it before rank 0 raises
                fatal = e                                # (the others would sit in the broadcast until the collective times out)
        if multi:
            box = [chosen, None if fatal is None else repr(fatal)]
            dist.broadcast_object_list(box, src=0)
            chosen, why = box
            if why is not None:
                raise RuntimeError(f"rank {self._rank()}: rank 0 could not choose a checkpoint: {why}") from fatal
        elif fatal is not None:
            raise fatal
        ok, err = 1, None
        if chosen is not None:
            try:
                if payload is None:
                    payload = self._read_npz(chosen)
                self._apply_npz(*payload)
            except Exception as e:                      # reported after the ranks have compared notes, never swallowed
                ok, err = 0, e
        if multi:
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                raise RuntimeError(f"rank {self._rank()}: the ranks disagree about checkpoint {chosen} (this rank: "
                                   f"{'loaded' if ok else repr(err)}); refusing to train on diverged replicas") from err
        elif err is not None:
            raise err
        return chosen

    def _save_checkpoint(self, max_to_keep=3):
        """tf.train.CheckpointManager(ckpt, checkpoint_dir, max_to_keep=3).save() (SHM.py:944, 1127).  Rank 0 writes (to a
        temporary name, then an atomic rename) and prunes; every rank leaves through a barrier, so nobody races ahead into a
        collective while the file is still being written and nobody deletes a file another rank is about to open."""
        import os
        path = None
        self._check_abort(sync=True)     # never write a checkpoint behind a step whose kernels gave up
        if self._rank() == 0:
            c = self._checkpoints()
            n = int(os.path.basename(c[-1])[5:-4]) + 1 if c else 1
            path = os.path.join(self.checkpoint_save_dir, f"ckpt-{n}.npz")
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:                 # a file object: np.savez would append ".npz" to a name
                self.save_npz(f)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, path)
            for old in (c + [path])[:-max_to_keep]:
                os.remove(old)
        self._barrier()
        return path

    # ------------------------------------------------------------------ inference (test.py:218-297)
    def infer(self, rgb):
        """Forward-only path of the reference's evaluation script (/root/reference/test.py:218-297):
        standardised YUV of ONE rgb image -> G once with only view 0 populated (target ED) -> the
        generated RGB, whose channel 0 feeds five cyclic G calls.  rgb [B,S,S,3] in [0,1].
        Returns (gen_rgb [B,S,S,3], [5 x cyc_rgb [B,S,S,3]]); sets the same attributes as test.py."""
        if self.G is None:
            self.build()
        G, A = self.G, self.arena
        S = self.image_size
        x = self._dev(rgb)
        B = x.shape[0]
        npix = S * S
        G.prepare_weights()
        yuv, scale = self.preprocess(x, "inf")
        if self.SpecSeg is None:
            self.SpecSeg = self.build_specseg()
        self.specular_candidate = self.SpecSeg.forward_plane(yuv, 3, 0, B, tag="specseg/inf")   # test.py:221
        cbcr = yuv[..., 1:].contiguous()                       # averageCbCr = the input's own CbCr (test.py:224)
        ys = [yuv] * 5
        adt, PAD_C = self.compute_dtype, self.pad
        gen_in = A.get("inf/in", (B, S, S, PAD_C), adt)
        ops.build_gen_input(ys, None, 0b11110, 0, gen_in, B, npix)          # views 1..4 zero, one-hot = ED
        attn = G.attention_forward(self.specular_candidate, B) if self.attention == "live" else None
        gen_Y = G.forward(gen_in, "inf1", attn=attn)
        gen_rgb = A.get("inf/rgb", (B, S, S, 3))
        ops.yuv2rgb(gen_Y, cbcr, None, gen_rgb, None, B, B, npix)
        orig_Ych = gen_rgb[..., 0:1].contiguous()              # test.py:252
        cyc_in = A.get("inf/cyc_in", (5 * B, S, S, PAD_C), adt)
        ops.build_gen_input(ys, orig_Ych, 0b11111, 1, cyc_in, B, npix)      # view k zero, the others = orig_Ych
        cyc_Y = G.forward(cyc_in, "inf5", attn=attn, parts=self.forward_parts)
        cyc_rgb = A.get("inf/cyc_rgb", (5 * B, S, S, 3))
        ops.yuv2rgb(cyc_Y, cbcr, None, cyc_rgb, None, 5 * B, B, npix)
        self.gen_input, self.gen_Y, self.gen_rgb = gen_in, gen_Y, gen_rgb
        self.stddev_arr = [scale]
        outs = [cyc_rgb[k * B:(k + 1) * B] for k in range(5)]
        (self.cyc_gen0_rgb, self.cyc_gen45_rgb, self.cyc_gen90_rgb, self.cyc_gen135_rgb, self.cyc_genED_rgb) = outs
        return gen_rgb, outs

    # ------------------------------------------------------------------ weights interchange
    def save_npz(self, path):
        """Weights + IN betas + Adam state in a neutral .npz (Keras variable order and layouts)."""
        d = {}
        for name, M in (("G", self.G), ("D", self.D)):
            for i, w in enumerate(M.get_weights()):
                d[f"{name}/var{i:02d}"] = w
            for i, b in enumerate(M.betas):
                d[f"{name}/beta{i:02d}"] = b.cpu().numpy()
            d[f"{name}/adam_m"] = M.P.m.cpu().numpy()
            d[f"{name}/adam_v"] = M.P.v.cpu().numpy()
            d[f"{name}/iterations"] = np.int64(M.P.iterations)
        if self.SpecSeg is not None:
            for i, w in enumerate(self.SpecSeg.get_weights()):
                d[f"SpecSeg/var{i:02d}"] = w
        # trainer state a resumed run continues from: the step-level draw streams (flags / TARGET_LABELS generator, the
        # Philox counter of the noise and dropout kernels), the epoch, and which graph the variables belong to
        import json
        d["trainer/state"] = np.array(json.dumps({
            "attention": self.attention, "epoch": int(self.epoch), "draw_count": int(self._draw_count),
            "rng": self._rng.bit_generator.state, "batch_step": int(getattr(self, "batch_step", 0))}))
        np.savez(path, **d)

    def _read_npz(self, path):
        """Read and validate a save_npz file completely without mutating anything: every array this trainer's models need is
        decoded (zipfile checks the CRC of each member as it is read) and shape-checked.  Returns (arrays, trainer state)."""
        import json
        if self.G is None:
            self.build()
        with np.load(path) as z:
            state = None
            if "trainer/state" in z.files:
                state = json.loads(str(z["trainer/state"]))
                if state["attention"] != self.attention:
                    raise KeyError(f"checkpoint {path} holds an attention='{state['attention']}' model, this trainer was built with "
                                   f"attention='{self.attention}' (the live branch adds 20 + 4 variables)")
            arrays = {}
            for name, M in (("G", self.G), ("D", self.D)):
                for i, shape in enumerate(M.P.shapes[:len(M.P.vars)]):
                    a = z[f"{name}/var{i:02d}"]
                    if tuple(a.shape) != tuple(shape):
                        raise KeyError(f"checkpoint {path}: {name}/var{i:02d} has shape {tuple(a.shape)}, this model's is {tuple(shape)}")
                    arrays[f"{name}/var{i:02d}"] = a
                for i in range(len(M.betas)):
                    arrays[f"{name}/beta{i:02d}"] = z[f"{name}/beta{i:02d}"]
                for k in ("adam_m", "adam_v"):
                    a = z[f"{name}/{k}"]
                    if a.size != M.P.m.numel():
                        raise KeyError(f"checkpoint {path}: {name}/{k} has {a.size} elements, this model's flat buffer {M.P.m.numel()}")
                    arrays[f"{name}/{k}"] = a
                arrays[f"{name}/iterations"] = z[f"{name}/iterations"]
            if "SpecSeg/var00" in z.files:
                if self.SpecSeg is None:
                    self.SpecSeg = self.build_specseg()
                for i in range(len(self.SpecSeg.vars)):
                    arrays[f"SpecSeg/var{i:02d}"] = z[f"SpecSeg/var{i:02d}"]
        return arrays, state

    def _apply_npz(self, z, state):
        for name, M in (("G", self.G), ("D", self.D)):
            M.set_weights([z[f"{name}/var{i:02d}"] for i in range(len(M.P.vars))])
            M.set_betas([z[f"{name}/beta{i:02d}"] for i in range(len(M.betas))])
            M.P.m.copy_(torch.from_numpy(z[f"{name}/adam_m"]))
            M.P.v.copy_(torch.from_numpy(z[f"{name}/adam_v"]))
            M.P.iterations = int(z[f"{name}/iterations"])
        if "SpecSeg/var00" in z:
            self.SpecSeg.set_weights([z[f"SpecSeg/var{i:02d}"] for i in range(len(self.SpecSeg.vars))])
        if state is not None:            # continue the draw streams instead of replaying the first run's opening steps
            self.epoch = int(state["epoch"])
            self._draw_count = int(state["draw_count"])
            self._rng.bit_generator.state = state["rng"]

    def load_npz(self, path):
        """Inverse of save_npz.  The file is read and validated as a whole first (`_read_npz`), so a damaged or mismatching
        file raises before any weight, Adam moment or draw-stream state has been touched."""
        self._apply_npz(*self._read_npz(path))

    def _img_ws(self, B):
        n = ops.image_losses_workspace(B, self.image_size)
        return self.arena.get("loss/ws", ((n + 3) // 4,))

    # ------------------------------------------------------------------ named losses (lazy D2H)
    def losses(self):
        """Compose the reference's named loss scalars (mean over the batch) from the two f64 loss
        vectors the kernels filled.  Synchronises (device->host copy) -- call it off the hot path."""
        if self._loss_cache is not None:
            return self._loss_cache
        L = self._last
        self._check_abort(sync=True)
        d = (L.dl.cpu().numpy() / L.B).tolist()
        i = (L.il.cpu().numpy() / L.B).tolist()
        D1_RF, D3_RF = d[0], d[1]
        D2_RF = d[4] + d[2]
        D4_RF = d[5] + d[3] + D2_RF
        D1_cls, D3_cls, D4_cls = d[6], d[7], d[8]
        L1 = (i[1] + i[2] + i[3] + i[4] + i[0]) / 5.0 + i[5] * 10.0
        ssim_loss = (i[11] + i[12] + i[13] + i[14] + i[15] * 10.0) / 5.0
        content, style = i[16], i[17]
        nst = 100.0 * style + content
        sp = (L.sl.cpu().numpy() / (L.B * L.npix * 3.0)).tolist()       # reduce_mean over [B,S,S,3]
        out = {
            "total_Generator_loss": (D1_RF + D3_RF) / 6.0 + 10.0 * L1 + 10.0 * ssim_loss + 10.0 * nst,
            "total_Discriminator_loss": (D1_cls + D3_cls) / 6.0 + (D2_RF + D4_RF) / 6.0 + 0.5 * D4_cls + 10.0 * nst,
            "total_Classification_loss": (D4_cls + nst) * 10.0,
            "G_gan_loss": (D3_RF + D1_RF) / 6.0, "G_clsf_loss": (D3_cls + D1_cls) / 6.0,
            "D1_RealFake_loss": D1_RF, "D3_RealFake_cyc": D3_RF, "D2_RealFake_target": D2_RF,
            "D4_RealFake_cyc": D4_RF, "D1_classification_loss": D1_cls, "D3_classification_loss": D3_cls,
            "D4_classification_loss": D4_cls, "L1_loss_Gen": L1, "ssim_cyc_loss": ssim_loss,
            "content_loss": content, "style_loss": style, "total_NST_loss": nst,
            "Spec_loss": (sp[0] + sp[1] + sp[2] + sp[3]) / 5.0 + sp[4] * 5.0,
            "ssim": [i[6 + k] for k in range(5)],
        }
        self._loss_cache = out
        return out

    def __getattr__(self, name):
        # reference attribute names for the loss scalars (self.total_Generator_loss, ...)
        if name in LOSS_NAMES and self.__dict__.get("_last") is not None:
            return self.losses()[name]
        raise AttributeError(name)

    @property
    def gen_rgb_output(self):
        """SHM.py:548-550: yuv_to_rgb(gen_YCbCr * mean(stddev_arr) * 255) (display only).  Plain torch tensor arithmetic on purpose: like
        `custom_per_image_standardization` and `gram_matrix` this is an API-parity helper OFF the hot path (nothing in `train_step` / `infer`
        calls it) -- the three are the only torch compute in the package; the step itself runs on the library's kernels alone."""
        L = self._last
        avg = torch.stack([s.mean() for s in self.stddev_arr]).mean()
        yuv = torch.cat([self.gen_Y, L.cbcr], dim=3) * avg * 255.0
        k = torch.tensor([[1.0, 1.0, 1.0], [0.0, -0.394642334, 2.03206185], [1.13988303, -0.58062185, 0.0]],
                         device=yuv.device)
        return yuv @ k
