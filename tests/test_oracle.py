"""CPU tests of the oracle: structural known answers from the reference's committed Keras
summaries, two independent restatements against each other (NumPy explicit vs PyTorch autograd),
finite differences, and the committed golden fixtures."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import step_torch as st
from oracle import tf_ops_np as tn

GOLD = Path(__file__).resolve().parent / "golden"

# Known answers transcribed from /root/reference/Generator_summary.txt and
# Discriminator_summary.txt (Keras .summary() at image_size 128, filter_size 64):
# (layer name, output H, output C, param count)
G_KAT = [
    ("conv2d", 128, 64, 5824), ("conv2d_1", 128, 64, 36928), ("conv2d_4", 64, 128, 73856),
    ("conv2d_5", 64, 128, 147584), ("conv2d_8", 32, 256, 295168), ("conv2d_9", 32, 256, 590080),
    ("conv2d_12", 16, 512, 1180160), ("conv2d_13", 16, 512, 2359808), ("conv2d_16", 8, 512, 262656),
    ("conv2d_17", 8, 512, 262656), ("conv2d_transpose", 16, 512, 2359808), ("conv2d_18", 16, 512, 4719104),
    ("conv2d_19", 16, 512, 2359808), ("conv2d_transpose_1", 32, 256, 1179904), ("conv2d_20", 32, 256, 1179904),
    ("conv2d_21", 32, 256, 590080), ("conv2d_transpose_2", 64, 128, 295040), ("conv2d_22", 64, 128, 295040),
    ("conv2d_23", 64, 128, 147584), ("conv2d_transpose_3", 128, 64, 73792), ("conv2d_24", 128, 64, 73792),
    ("conv2d_25", 128, 64, 36928), ("conv2d_26", 128, 1, 65),
]
D_KAT = [("conv2d_27", 1728), ("conv2d_28", 73728), ("conv2d_29", 294912), ("conv2d_30", 1179648),
         ("conv2d_33", 4718592), ("conv2d_34", 9216), ("dense", 81920)]


def test_known_answers_param_counts():
    spec = st.generator_spec(64)
    shapes = st.generator_var_shapes(64)
    assert [s[0] for s in spec] == [k[0] for k in G_KAT]
    for i, (name, h, c, cnt) in enumerate(G_KAT):
        assert int(np.prod(shapes[2 * i])) + int(np.prod(shapes[2 * i + 1])) == cnt, name
        assert spec[i][4] == c
    assert sum(int(np.prod(s)) for s in shapes) == 18525569          # Generator_summary.txt:621
    d128 = st.discriminator_spec(64, 128)
    assert [(n, int(np.prod(s))) for n, _, s in d128] == D_KAT
    assert sum(int(np.prod(s)) for _, _, s in d128) == 6359744       # Discriminator_summary.txt:179
    assert sum(int(np.prod(s)) for _, _, s in st.discriminator_spec(64, 256)) == 6605504
    assert len(st.generator_in_channels(64)) == 18 and len(st.discriminator_in_channels(64)) == 5


def test_generator_output_shapes():
    F, S = 16, 32
    g, d, gb, db = st.init_params(F, 64)
    x = torch.zeros(1, S, S, 10, dtype=torch.float64)
    rec = []
    y = st.generator_forward([torch.from_numpy(a).double() for a in g], [torch.from_numpy(b).double() for b in gb], x, F, record=rec)
    assert tuple(y.shape) == (1, S, S, 1)
    hs = [z.shape[2] for z, _ in rec]
    assert hs == [32, 32, 16, 16, 8, 8, 4, 4, 2, 2, 4, 4, 8, 8, 16, 16, 32, 32]      # x S/128 of G_KAT


def test_same_padding_rule():
    assert tn.same_pads(256, 3, 1) == (256, 1, 1)
    assert tn.same_pads(256, 3, 2) == (128, 0, 1)        # asymmetric: NOT PyTorch's symmetric pad 1
    assert tn.same_pads(7, 3, 2) == (4, 1, 1)
    assert tn.same_pads(8, 1, 1) == (8, 0, 0)


@pytest.mark.parametrize("stride", [1, 2])
def test_conv_numpy_vs_torch(stride):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 6, 5))
    w = rng.standard_normal((3, 3, 5, 7))
    a = tn.conv2d_same(x, w, stride)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    b = st.conv2d_same(xt, wt, stride)
    assert np.abs(a - b.detach().permute(0, 2, 3, 1).numpy()).max() < 1e-12
    dy = rng.standard_normal(a.shape)
    dx, dw = tn.conv2d_same_bwd(x, w, dy, stride)
    gx, gw = torch.autograd.grad(b, [xt, wt], torch.from_numpy(dy).permute(0, 3, 1, 2))
    assert np.abs(dx - gx.permute(0, 2, 3, 1).numpy()).max() < 1e-12
    assert np.abs(dw - gw.numpy()).max() < 1e-11


def test_conv_transpose_is_gradient_of_strided_conv():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 4, 5, 3))
    w = rng.standard_normal((3, 3, 6, 3))            # Keras [kh,kw,Cout,Cin]
    a = tn.conv2d_transpose_same(x, w)
    b = st.conv2d_transpose_same(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w)).permute(0, 2, 3, 1).numpy()
    assert a.shape == (2, 8, 10, 6) and np.abs(a - b).max() < 1e-12
    big = torch.zeros(2, 6, 8, 10, dtype=torch.float64, requires_grad=True)
    y = st.conv2d_same(big, torch.from_numpy(w), 2)      # w read as HWIO with I=6, O=3
    g, = torch.autograd.grad(y, big, torch.from_numpy(x).permute(0, 3, 1, 2))
    assert np.abs(g.permute(0, 2, 3, 1).numpy() - a).max() < 1e-12
    # a symmetric-pad-1 stride-2 conv is NOT the same operator
    y2 = torch.nn.functional.conv2d(big, torch.from_numpy(w).permute(3, 2, 0, 1), stride=2, padding=1)
    g2, = torch.autograd.grad(y2, big, torch.from_numpy(x).permute(0, 3, 1, 2))
    assert np.abs(g2.permute(0, 2, 3, 1).numpy() - a).max() > 1e-3


def test_instance_norm_and_pool_and_ssim():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 6, 6, 4)) * 3 + 1
    beta = rng.standard_normal(4)
    a = tn.instance_norm(x, beta)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
    b = st.instance_norm(xt, torch.from_numpy(beta))
    assert np.abs(a - b.detach().permute(0, 2, 3, 1).numpy()).max() < 1e-12
    # normalised output: mean = beta, variance = var/(var+eps)
    assert np.abs(a.mean(axis=(1, 2)) - beta).max() < 1e-12
    dy = rng.standard_normal(x.shape)
    g, = torch.autograd.grad(b, xt, torch.from_numpy(dy).permute(0, 3, 1, 2))
    assert np.abs(tn.instance_norm_bwd(x, dy) - g.permute(0, 2, 3, 1).numpy()).max() < 1e-10
    p = tn.avg_pool2(x)
    assert np.abs(p - torch.nn.functional.avg_pool2d(xt, 2).detach().permute(0, 2, 3, 1).numpy()).max() < 1e-12
    u, v = rng.random((2, 16, 16, 3)), rng.random((2, 16, 16, 3))
    assert np.abs(tn.ssim(u, v) - st.ssim(torch.from_numpy(u), torch.from_numpy(v)).numpy()).max() < 1e-12
    assert abs(float(tn.ssim(u, u)[0]) - 1.0) < 1e-12
    assert abs(tn.gauss_window().sum() - 1.0) < 1e-12


@pytest.mark.parametrize("T", [1.0, 0.83, 1.17])
def test_xent_executed_vs_intended_gradient(T):
    """tf.nn.softmax_cross_entropy_with_logits as executed: values equal the plain formula; the gradient is
    grad_loss * (softmax - labels).  For a label row [0,0,0,0,T] that coincides with the true derivative at T = 1 and
    differs from it by (T - 1) * softmax otherwise (SHM.py:477, 688, 702); one-hot rows never differ."""
    rng = np.random.default_rng(11)
    lg = rng.standard_normal((4, 5))
    lab = np.zeros((4, 5)); lab[:, 4] = T
    gl = rng.standard_normal(4)
    grads = {}
    for mode in ("executed", "intended"):
        z = torch.from_numpy(lg).requires_grad_(True)
        v = st.softmax_xent(z, torch.from_numpy(lab), mode)
        assert np.abs(v.detach().numpy() - tn.softmax_xent(lab, lg)).max() < 1e-13
        grads[mode], = torch.autograd.grad((v * torch.from_numpy(gl)).sum(), [z])
    p = torch.softmax(torch.from_numpy(lg), -1).numpy()
    assert np.abs(grads["executed"].numpy() - gl[:, None] * tn.softmax_xent_backprop(lab, lg)).max() < 1e-13
    assert np.abs(grads["intended"].numpy() - gl[:, None] * (T * p - lab)).max() < 1e-13
    d = (grads["intended"] - grads["executed"]).numpy()
    assert np.abs(d - gl[:, None] * (T - 1.0) * p).max() < 1e-13
    if T == 1.0:
        assert np.abs(d).max() < 1e-15


def test_step_xent_mode_only_moves_the_discriminator_class_path():
    """Whole step, both modes: every loss value and the whole generator gradient are identical (G receives no
    classification gradient); the discriminator gradient differs (through D1's class logits) when T != 1."""
    S, F, B = 32, 8, 1
    g, d, gb, db = st.init_params(F, S)
    inp = st.make_inputs(B, S)
    dr = st.make_draws(0, B, S, F)
    dr.target_label = 1.15
    sf = st.style_factor_intended(S)
    a = st.train_step(g, d, gb, db, inp, dr, sf, F, xent_mode="executed")
    b = st.train_step(g, d, gb, db, inp, dr, sf, F, xent_mode="intended")
    assert a["losses"] == b["losses"]
    assert all(torch.equal(x, y) for x, y in zip(a["gG"], b["gG"]))
    rel = float((a["gD"][6] - b["gD"][6]).norm() / b["gD"][6].norm())
    assert rel > 1e-3, rel                      # the Dense(5) kernel sees the (T-1)*softmax/6 difference
    dr.target_label = 1.0
    a = st.train_step(g, d, gb, db, inp, dr, sf, F, xent_mode="executed")
    b = st.train_step(g, d, gb, db, inp, dr, sf, F, xent_mode="intended")
    assert all(float((x - y).abs().max()) < 1e-14 for x, y in zip(a["gD"], b["gD"]))


def test_colour_standardise_rescale_gram_xent_adam():
    rng = np.random.default_rng(3)
    x = rng.random((2, 8, 8, 3))
    yuv = tn.rgb_to_yuv(x)
    assert np.abs(tn.yuv_to_rgb(yuv) - x).max() < 1e-6           # TF's two matrices are inverse to ~1e-7
    assert np.abs(yuv - st.rgb_to_yuv(torch.from_numpy(x)).numpy()).max() < 1e-14
    s_np, sc_np = tn.per_image_standardization(yuv)
    s_t, sc_t = st.per_image_standardization(torch.from_numpy(yuv))
    assert np.abs(s_np - s_t.numpy()).max() < 1e-12 and np.abs(sc_np - sc_t.numpy()).max() < 1e-14
    flat = np.full((1, 4, 4, 3), 0.5)
    assert tn.per_image_standardization(flat)[1][0] == 1.0 / 256.0     # min_stddev clamp (SHM.py:1293)
    r = tn.rescale_01(yuv)
    assert r.min() == 0.0 and r.max() == 1.0
    assert np.abs(r - st.rescale_01(torch.from_numpy(yuv)).numpy()).max() < 1e-14
    assert np.abs(tn.rescale_01(flat)).max() == 0.0                     # divide_no_nan
    assert np.abs(tn.gram_matrix(yuv) - st.gram_matrix(torch.from_numpy(yuv)).numpy()).max() < 1e-14
    lg = rng.standard_normal((3, 5))
    lab = np.zeros((3, 5)); lab[:, 4] = 1.1
    ref = -1.1 * torch.log_softmax(torch.from_numpy(lg), -1)[:, 4].numpy()
    assert np.abs(tn.softmax_xent(lab, lg) - ref).max() < 1e-12
    # Adam: oracle list form == numpy scalar form
    w, g = rng.standard_normal(50), rng.standard_normal(50) * 3
    wt = [torch.from_numpy(w.copy())]
    stt = st.AdamState([torch.zeros(50, dtype=torch.float64)], [torch.zeros(50, dtype=torch.float64)], iterations=7)
    st.adam_apply(wt, [torch.from_numpy(g)], stt, 2e-5, 0.5, 0.99)
    w2, m2, v2 = tn.adam_update(w, np.zeros(50), np.zeros(50), g, 7, 2e-5, 0.5, 0.99)
    assert np.abs(wt[0].numpy() - w2).max() < 1e-15 and stt.iterations == 8


def test_style_factor_int32_wrap():
    """SURVEY finding 7: tf.math.square(2*9*S*S) on a Python int is an int32 op."""
    assert st.style_factor_as_executed(256) == float("inf")
    assert abs(st.style_factor_as_executed(128) - 1.0 / 2 ** 30) < 1e-18
    assert abs(st.style_factor_intended(128) - 1.0 / (2 * 9 * 128 * 128) ** 2) < 1e-30


def test_train_step_gradient_finite_difference():
    """Directional finite difference of total_G and total_D+total_C through the oracle itself."""
    S, F, B = 64, 16, 1
    g, d, gb, db = st.init_params(F, S)
    inp = st.make_inputs(B, S)
    dr = st.make_draws(0, B, S, F)
    sf = st.style_factor_intended(S)
    r = st.train_step(g, d, gb, db, inp, dr, sf, F)
    rng = np.random.default_rng(4)
    eps = 1e-5
    for which in ("G", "D"):
        vars_ = g if which == "G" else d
        dirs = [rng.standard_normal(v.shape) * (np.abs(v).max() + 1e-3) for v in vars_]
        plus = [v + eps * u for v, u in zip(vars_, dirs)]
        minus = [v - eps * u for v, u in zip(vars_, dirs)]
        if which == "G":
            lp = st.train_step(plus, d, gb, db, inp, dr, sf, F, need_grads=False)["losses"]["total_Generator_loss"]
            lm = st.train_step(minus, d, gb, db, inp, dr, sf, F, need_grads=False)["losses"]["total_Generator_loss"]
            an = sum(float((gr * torch.from_numpy(u)).sum()) for gr, u in zip(r["gG"], dirs))
        else:
            f = lambda L: L["total_Discriminator_loss"] + L["total_Classification_loss"]
            lp = f(st.train_step(g, plus, gb, db, inp, dr, sf, F, need_grads=False)["losses"])
            lm = f(st.train_step(g, minus, gb, db, inp, dr, sf, F, need_grads=False)["losses"])
            an = sum(float((gr * torch.from_numpy(u)).sum()) for gr, u in zip(r["gD"], dirs))
        fd = (lp - lm) / (2 * eps)
        # the loss has kinks (|.| in L1, LeakyReLU, min/max): a finite step crosses a few of them
        assert abs(fd - an) <= 1e-2 * max(abs(an), 1e-6), (which, fd, an)


def test_batch_rule_is_mean_of_single_sample_steps():
    """a B=2 step == mean of two independent B=1 steps (losses and gradients)."""
    S, F = 64, 16
    g, d, gb, db = st.init_params(F, S)
    inp = st.make_inputs(2, S)
    dr = st.make_draws(1, 2, S, F)
    sf = st.style_factor_intended(S)
    r2 = st.train_step(g, d, gb, db, inp, dr, sf, F)
    acc = None
    for b in range(2):
        drb = st.StepDraws(dr.flags, dr.target_label, dr.noise[[b, 2 + b]], dr.keep_mask[[b, 2 + b]])
        rb = st.train_step(g, d, gb, db, [a[b:b + 1] for a in inp], drb, sf, F)
        gs = [t / 2 for t in rb["gG"] + rb["gD"]]
        acc = gs if acc is None else [a + t for a, t in zip(acc, gs)]
    for a, t in zip(acc, r2["gG"] + r2["gD"]):
        assert float((a - t).abs().max()) <= 1e-10 * max(1.0, float(t.abs().max()))


@pytest.mark.parametrize("name", ["step_S64_F16_B1.npz", "step_S64_F16_B2.npz"])
def test_golden_step_fixture(name):
    gold = np.load(GOLD / name)
    S, F, B, step = [int(v) for v in gold["meta"]]
    g, d, gb, db = st.init_params(F, S)
    from oracle import specseg_torch as sp
    r = st.train_step(g, d, gb, db, st.make_inputs(B, S), st.make_draws(step, B, S, F), st.style_factor_intended(S), F,
                      specseg=sp.init_specseg(seed=44 + step))
    assert "Spec_loss" in r["losses"]
    assert np.abs(r["outs"]["specular_candidate"].numpy() - gold["specular_candidate"]).max() < 1e-6
    for k, v in r["losses"].items():
        assert abs(v - float(gold[f"loss/{k}"])) <= 1e-9 * max(1.0, abs(v)), k
    assert np.abs(r["outs"]["gen_Y"].numpy() - gold["gen_Y"]).max() < 1e-6
    for nm, gr in (("gG", r["gG"]), ("gD", r["gD"])):
        n = np.array([float(t.norm()) for t in gr])
        assert np.abs(n - gold[f"{nm}/norm"]).max() <= 1e-8 * max(1.0, n.max())


# ------------------------------------------------------------------ SpecSeg restatement (S1, L5)
def test_specseg_param_count_known_answer():
    """/root/reference/SpecSeg_summary.txt:118-120: 1,942,801 params, 992 non-trainable; per-layer
    counts from the same file."""
    from oracle import specseg_torch as sp
    spec = sp.specseg_spec()
    n = sum(int(np.prod(s)) for _, s in spec)
    nt = sum(int(np.prod(s)) for k, s in spec if k in ("bn_mean", "bn_var"))
    assert (n, nt) == (1942801, 992)
    # conv2d 160, conv2d_1 2320, conv2d_9 590080, conv2d_transpose 131200, conv2d_10 295040, conv2d_18 17
    pair = lambda i: int(np.prod(spec[i][1])) + int(np.prod(spec[i + 1][1]))
    assert pair(0) == 160 and pair(2) == 2320 and pair(34) == 590080
    assert pair(40) == 131200 and pair(42) == 295040 and pair(64) == 17
    from shmgan_amd.specseg import specseg_variables
    assert [tuple(s) for _, _, s in specseg_variables()] == [tuple(s) for _, s in spec]


def test_specseg_oracle_forward_properties():
    from oracle import specseg_torch as sp
    import torch
    w = sp.init_specseg(seed=1)
    x = np.random.default_rng(0).standard_normal((2, 32, 32, 1))
    m = sp.specseg_forward(w, x)
    assert m.shape == (2, 32, 32, 1) and float(m.min()) > 0 and float(m.max()) < 1
    # samples are independent in inference mode
    m0 = sp.specseg_forward(w, x[:1])
    assert np.allclose(m[:1].numpy(), m0.numpy(), atol=1e-12)
    # float32 evaluation of the same restatement agrees
    m32 = sp.specseg_forward(w, x, dtype=torch.float32)
    assert np.abs(m32.numpy() - m.numpy()).max() < 1e-5


def test_specseg_golden_fixture():
    from oracle import specseg_torch as sp
    gold = np.load(GOLD / "specseg_S32.npz")
    m = sp.specseg_forward(sp.init_specseg(seed=3), gold["x"])
    assert np.abs(m.numpy() - gold["mask"]).max() < 1e-12


# ------------------------------------------------------------------ input pipeline restatement (N4)
def test_resize_bilinear_known_answers():
    """Hand-computed values of ResizeBilinear(half_pixel_centers=True): 2x2 -> 4x4 upsampling puts the sample
    points at -0.25, 0.25, 0.75, 1.25 source pixels (clamped at the border), 4 -> 2 averages pairs."""
    from oracle import data_np as dn
    a = np.array([[0.0, 4.0], [8.0, 12.0]])[..., None]
    out = dn.resize_bilinear(a, 4, 4)[..., 0]
    assert np.allclose(out[0], [0.0, 1.0, 3.0, 4.0])
    assert np.allclose(out[:, 0], [0.0, 2.0, 6.0, 8.0])
    assert np.allclose(out[1, 1], 3.0) and np.allclose(out[3, 3], 12.0)
    b = np.arange(4.0)[None, :, None]
    assert np.allclose(dn.resize_bilinear(b, 1, 2)[0, :, 0], [0.5, 2.5])
    c = np.random.default_rng(0).integers(0, 256, (7, 5, 3)).astype(np.uint8)
    assert np.array_equal(dn.resize_bilinear(c, 7, 5), c.astype(np.float32))          # identity at equal size
    v = dn.load_view(c, 8)
    assert v.shape == (8, 8, 3) and v.min() >= 0 and v.max() <= 1
    assert np.array_equal(dn.load_view(c, 8, flip_ud=False)[::-1], v)


def test_dataset_listing_is_sorted_and_filtered(tmp_path):
    """image_dataset_from_directory(shuffle=False) order: sorted file names, image extensions only."""
    from shmgan_amd.data import list_images
    for name in ("b_10.png", "a_2.PNG", "a_10.jpg", "notes.txt", "c.bmp"):
        (tmp_path / name).write_bytes(b"x")
    assert [p.split("/")[-1] for p in list_images(tmp_path)] == ["a_10.jpg", "a_2.PNG", "b_10.png", "c.bmp"]


def test_live_attention_oracle_properties():
    """oracle.step_torch.train_step(attention=...): zero attention variables reproduce the executed graph exactly; the
    attention gradients agree with central finite differences of the generator / discriminator objectives."""
    from oracle import specseg_torch as sp
    S, F, B = 64, 8, 1
    g, d, gb, db = st.init_params(F, S)
    inp, dr, sf = st.make_inputs(B, S), st.make_draws(0, B, S, F), st.style_factor_intended(S)
    sw = sp.init_specseg(seed=44)
    att = st.init_attention(F, bias_std=0.05)
    base = st.train_step(g, d, gb, db, inp, dr, sf, F, specseg=sw, need_grads=False)
    zero = {k: [np.zeros_like(a) for a in v] for k, v in att.items()}
    rz = st.train_step(g, d, gb, db, inp, dr, sf, F, specseg=sw, attention=zero, need_grads=False)
    assert all(rz["losses"][k] == base["losses"][k] for k in base["losses"])
    r = st.train_step(g, d, gb, db, inp, dr, sf, F, specseg=sw, attention=att)
    assert len(r["gGa"]) == 16 and len(r["gDa"]) == 4

    def objective(a, which):
        out = st.train_step(g, d, gb, db, inp, dr, sf, F, specseg=sw, attention=a, need_grads=False)["losses"]
        return out["total_Generator_loss"] if which == "G" else out["total_Discriminator_loss"] + out["total_Classification_loss"]

    rng = np.random.default_rng(0)
    eps = 1e-5
    for which, key, grads in (("G", "G", r["gGa"]), ("D", "D", r["gDa"])):
        dirs = [rng.standard_normal(v.shape) * (np.abs(v).max() + 1e-3) for v in att[key]]
        ap = {k: [x.astype(np.float64).copy() for x in v] for k, v in att.items()}
        am = {k: [x.astype(np.float64).copy() for x in v] for k, v in att.items()}
        for i, u in enumerate(dirs):
            ap[key][i] += eps * u
            am[key][i] -= eps * u
        fd = (objective(ap, which) - objective(am, which)) / (2 * eps)
        an = sum(float((gr.numpy() * u).sum()) for gr, u in zip(grads, dirs))
        # the loss has kinks (|.| in L1, LeakyReLU, min/max): a finite step crosses a few of them
        assert abs(fd - an) <= 1e-2 * max(abs(an), 1e-6), (which, fd, an)


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds: zero counter / key and all-ones counter / key."""
    from oracle import rng_np
    z = rng_np.philox4x32_10([0, 0, 0, 0], [0, 0])
    assert [int(v) for v in z] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = rng_np.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)
    assert [int(v) for v in f] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    x = rng_np.randn(1 << 16, 0.1, 7, 0)
    assert abs(x.mean()) < 2e-3 and abs(x.std() - 0.1) < 2e-3
    assert abs(rng_np.keep_mask(1 << 16, 0.2, 7, 1).mean() - 0.8) < 6e-3
