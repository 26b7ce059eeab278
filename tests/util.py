"""Shared helpers for the parity tests (oracle side in float64 on the CPU)."""
import numpy as np
import torch

from oracle import step_torch as st


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def t64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def nchw(a):
    return t64(a).permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def conv_ref(x, w_hwio, stride):
    return nhwc(st.conv2d_same(nchw(x), t64(w_hwio), stride))


def pad_c(x, c):
    """zero-pad the channel axis to c."""
    out = np.zeros(x.shape[:-1] + (c,), x.dtype)
    out[..., :x.shape[-1]] = x
    return out


# ---------------------------------------------------------------------------------------------------------------------
# LeakyReLU kink pinning and gradient fixtures (whole-step tests on the device)

def pin_kinks(m, gold, tile=1):
    """Hook for trainer.before_backward: put the device on the float64 oracle's side of every LeakyReLU kink.  The
    fixture lists, per pass and layer, the pre-activations with |z| < kink/thr in float64 and their sign
    (oracle.step_torch.KinkRecorder); float32 rounding can only disagree about those.  Where the stored activation has
    the other sign it is replaced by +-1e-30 (a forward change of at most 0.8*thr at a few elements of tensors that are
    already consumed), so the backward pass differentiates the same piecewise-linear function as the oracle did.
    tile = R: the device runs the fixture's batch tiled R times (batch rule: B = R * B_fixture independent samples); the
    fixture's flat indices address [groups x B_fixture] rows and are mapped to the R copies of each row.
    Returns (hook, dict that receives {(tag, layer): (listed, flipped)})."""
    stats = {}
    b_fix = int(gold["meta"][2])

    def hook():
        tensors = {"g1": m.G.lrelu_tensors("g1"), "cyc": m.G.lrelu_tensors("cyc"), "d": m.D.lrelu_tensors()}
        for key in gold.files:
            if not key.startswith("kink/") or not key.endswith("/idx"):
                continue
            _, tag, layer, _ = key.split("/")
            idx = torch.from_numpy(gold[key]).cuda()
            pos = torch.from_numpy(gold[key[:-3] + "pos"]).cuda()
            flat = tensors[tag][int(layer)].view(-1)
            if tile > 1 and idx.numel():
                rows = flat.numel() // (tensors[tag][int(layer)].shape[0])           # elements per batch row
                row, off = idx // rows, idx % rows
                grp, b = row // b_fix, row % b_fix
                idx = torch.cat([((grp * tile + r) * b_fix + b) * rows + off for r in range(tile)])
                pos = pos.repeat(tile)
            assert idx.numel() == 0 or int(idx.max()) < flat.numel()
            cur = flat[idx]
            bad = (cur > 0) != pos
            # a disagreement may only concern a value float32 puts within rounding distance of zero
            assert float(cur[bad].abs().max()) < 10 * float(gold["kink/thr"]) if bool(bad.any()) else True
            flat[idx[bad]] = torch.where(pos[bad], 1e-30, -1e-30).to(flat.dtype)
            stats[(tag, int(layer))] = (int(idx.numel()), int(bad.sum()))
    return hook, stats


def pin_to_model(m, src):
    """Hook for trainer.before_backward of `m`: give every stored LeakyReLU output of `m` the SIGN it has in `src` (another
    trainer that has run the same forward passes on the same inputs, e.g. in bfloat16): where the signs differ the value
    becomes +-1e-30, so m's backward differentiates the piecewise-linear function src's backward differentiated.  This is
    the device-side twin of handing the float64 oracle the device's masks (oracle.step_torch `masks=`): it lets a bf16
    backward be compared with the fp32 backward of the same configuration at sizes where the float64 oracle cannot take
    the masks (committed fixtures) -- the fp32 backward itself is held to the oracle fixture by its own test.
    Returns (hook, dict that receives the fraction of elements whose sign differed per (tag, layer))."""
    stats = {}

    def hook():
        for tag, a, b in (("g1", m.G.lrelu_tensors("g1"), src.G.lrelu_tensors("g1")), ("cyc", m.G.lrelu_tensors("cyc"), src.G.lrelu_tensors("cyc")),
                          ("d", m.D.lrelu_tensors(), src.D.lrelu_tensors())):
            for li, (ta, tb) in enumerate(zip(a, b)):
                assert ta.shape == tb.shape, (tag, li, ta.shape, tb.shape)
                want = tb > 0
                bad = (ta > 0) != want
                stats[(tag, li)] = float(bad.float().mean())
                ta.copy_(torch.where(bad, torch.where(want, 1e-30, -1e-30).to(ta.dtype), ta))
    return hook, stats


def check_grad_fixture(m, gold, med_tol=1e-3, worst_tol=5e-2):
    """Per-tensor gradient norms and fixed random projections of the fixture (oracle/make_golden.py: one
    default_rng(99) stream over the G tensors, then the D tensors).  A LeakyReLU kink event (see
    test_train_step_parity: the fixture cannot pin the device's sign pattern) can move one layer by ~1e-2, so the
    worst tensor is held to 5e-2 of its norm and the median to 1e-3 -- unless the caller pinned the kinks (pin_kinks),
    which holds every tensor to worst_tol = 1e-3.  Returns {"gG" / "gD": (norm errors, projection errors)}."""
    rng = np.random.default_rng(99)
    out = {}
    for nm, P in (("gG", m.G.P), ("gD", m.D.P)):
        n = np.array([float(t.norm()) for t in P.grads])
        ref = gold[f"{nm}/norm"]
        ok = ref > 1e-12
        proj = np.array([float((t.detach().reshape(-1).double().cpu() * torch.from_numpy(rng.standard_normal(t.numel()))).sum())
                         for t in P.grads])
        err = np.abs(proj - gold[f"{nm}/proj"])[ok] / ref[ok]          # |<g - g_ref, r>| / |g_ref| ~ rel-L2 error
        nerr = np.abs(n[ok] / ref[ok] - 1)
        out[nm] = (nerr, err)
        assert nerr.max() < worst_tol, (nm, nerr.max(), int(nerr.argmax()))
        assert np.median(nerr) < med_tol
        assert err.max() < worst_tol, (nm, err.max(), int(err.argmax()))
        assert np.median(err) < med_tol, (nm, np.median(err))
    return out


def grad_cosines(m, ref, kernel_min=0.99, bias_min=0.97, model_min=0.995):
    """Per-tensor cosine between the weight gradients of two trainers on the device (bf16 gradients cosine
    >= 0.99 per tensor; bias vectors -- plain sums of dz, the part InstanceNorm's own backward cancels -- 0.97, as in
    test_bf16_train_step).  Returns the worst (name, index, cosine)."""
    worst = ("", -1, 1.0)
    for name, P, Q in (("D", m.D.P, ref.D.P), ("G", m.G.P, ref.G.P)):
        for i, (a, b) in enumerate(zip(P.grads, Q.grads)):
            bn = float(b.double().norm())
            if bn < 1e-12:
                continue
            cs = float((a.double() * b.double()).sum() / (a.double().norm() * bn))
            if cs < worst[2]:
                worst = (name, i, cs)
            assert cs > (kernel_min if b.dim() > 1 else bias_min), (name, i, cs)
        a, b = P.grad.double(), Q.grad.double()
        assert float((a * b).sum() / (a.norm() * b.norm())) > model_min, name
    return worst
