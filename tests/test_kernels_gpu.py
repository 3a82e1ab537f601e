"""Kernel-level parity on a real MI355X: each HIP convolution GEMM (forward,
data gradient, weight gradient, BN-statistics epilogue) against torch CPU fp32
ops on the same seeded inputs, called through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fedmlp_amd import spec

pytestmark = pytest.mark.gpu


def conv_names():
    names = ["conv1"]
    cin = 64
    for li, w in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"layer{li}.{b}"
            stride = 2 if (li > 1 and b == 0) else 1
            names += [p + ".conv1", p + ".conv2"]
            if stride != 1 or cin != w:
                names.append(p + ".downsample.0")
            cin = w
    return names


@pytest.fixture(scope="module")
def eng64():
    from fedmlp_amd.engine import Engine
    e = Engine("Resnet18", 5, 64, 64, 8)
    flat, cnt = spec.init_state("Resnet18", 5, 1037)
    e.set_state(flat, cnt)
    yield e, spec.flat_to_state_dict("Resnet18", 5, flat, cnt)
    e.close()


@pytest.fixture(scope="module")
def eng224():
    from fedmlp_amd.engine import Engine
    e = Engine("Resnet18", 5, 224, 224, 4)
    flat, cnt = spec.init_state("Resnet18", 5, 7)
    e.set_state(flat, cnt)
    yield e, spec.flat_to_state_dict("Resnet18", 5, flat, cnt)
    e.close()


@pytest.fixture(scope="module")
def eng224_forms():
    """one handle per product form of the conv GEMMs (fm_config.reserved[2]), same state"""
    from fedmlp_amd.engine import Engine
    flat, cnt = spec.init_state("Resnet18", 5, 7)
    engs = {}
    for sp in (0, 9, 6):
        engs[sp] = Engine("Resnet18", 5, 224, 224, 4, products=sp)
        engs[sp].set_state(flat, cnt)
    yield engs, spec.flat_to_state_dict("Resnet18", 5, flat, cnt)
    for e in engs.values():
        e.close()


def _nhwc(x_nchw, cpad):
    x = x_nchw.permute(0, 2, 3, 1).contiguous()
    if cpad > x.shape[3]:
        x = F.pad(x, (0, cpad - x.shape[3]))
    return x.contiguous()


def _check_conv(e, sd, ci, imgs, groups, seed):
    info = e.debug_conv_info(ci)
    name = conv_names()[ci]
    w = torch.from_numpy(sd[name + ".weight"])
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((imgs, info["cin"], info["hin"], info["win"]), generator=g)
    dy = torch.randn((imgs, info["cout"], info["hout"], info["wout"]), generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, info["stride"], info["pad"])
    y.backward(dy)
    dev = e.device
    x_d = _nhwc(x, info["cin_p"]).to(dev)
    dy_d = _nhwc(dy, info["cout"]).to(dev)
    # forward + BN partial statistics
    out = torch.empty((imgs, info["hout"], info["wout"], info["cout"]), device=dev)
    stats = torch.empty((groups, 2, info["cout"]), device=dev)
    e.debug_conv(0, ci, x_d, None, out, imgs, groups, stats)
    got = out.cpu().permute(0, 3, 1, 2)
    scale = y.detach().abs().max().item()
    np.testing.assert_allclose(got.numpy(), y.detach().numpy(), rtol=1e-4, atol=2e-5 * scale,
                               err_msg=f"fwd {name}")
    yg = y.detach().reshape(groups, imgs // groups, info["cout"], -1)
    s1 = yg.sum(dim=(1, 3)); s2 = (yg * yg).sum(dim=(1, 3))
    st = stats.cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), s1.numpy(), rtol=1e-3, atol=1e-3 * s2.sqrt().max().item())
    np.testing.assert_allclose(st[:, 1].numpy(), s2.numpy(), rtol=1e-4)
    # weight gradient (engine layout [cout][k][kw_p][cin_p])
    dw = torch.empty((info["cout"], info["Kw"]), device=dev)
    e.debug_conv(2, ci, x_d, dy_d, dw, imgs)
    want = wr.grad.permute(0, 2, 3, 1)                               # O,H,W,I
    want = F.pad(want, (0, info["cin_p"] - info["cin"], 0, info["kw_p"] - info["k"]))
    want = want.reshape(info["cout"], -1)
    want = F.pad(want, (0, info["Kw"] - want.shape[1]))              # packed stem: rows of 7*8*3 = 168 padded to 176
    scale = want.abs().max().item()
    np.testing.assert_allclose(dw.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5 * scale,
                               err_msg=f"wgrad {name}")
    # data gradient (not needed for the stem)
    if ci > 0:
        dx = torch.full((imgs, info["hin"], info["win"], info["cin"]), float("nan"), device=dev)
        e.debug_conv(1, ci, None, dy_d, dx, imgs)
        got = dx.cpu().permute(0, 3, 1, 2)
        want = xr.grad
        if info["k"] == 1 and info["stride"] == 2:
            # only parity class (0,0) is written by the 1x1 stride-2 dgrad; the block's 3x3
            # conv writes the rest (engine.hip backward_and_step)
            got, want = got[:, :, ::2, ::2], want[:, :, ::2, ::2]
        scale = want.abs().max().item()
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=2e-5 * scale,
                                   err_msg=f"dgrad {name}")


@pytest.mark.parametrize("ci", [0, 1, 6, 11, 16])
def test_split_products_are_fp32_accurate(eng224_forms, ci):
    """fm_config.reserved[2] = nine / six products (fp32 products as exact bf16 partial products on the bf16 matrix pipe,
    csrc/split3.h) against float64 convolutions: forward, data gradient and weight gradient are as close to float64 as the
    fp32-MFMA kernels are.  One handle per form."""
    engs, sd = eng224_forms
    e = engs[0]
    info = e.debug_conv_info(ci)
    imgs = 3
    g = torch.Generator().manual_seed(900 + ci)
    w = torch.from_numpy(sd[conv_names()[ci] + ".weight"]).double()
    x = torch.randn((imgs, info["cin"], info["hin"], info["win"]), generator=g)
    dy = torch.randn((imgs, info["cout"], info["hout"], info["wout"]), generator=g)
    xr = x.double().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, info["stride"], info["pad"])
    y.backward(dy.double())
    dev = e.device
    x_d = _nhwc(x, info["cin_p"]).to(dev)
    dy_d = _nhwc(dy, info["cout"]).to(dev)
    want_w = wr.grad.permute(0, 2, 3, 1)                               # O,H,W,I in the engine's padded layout
    want_w = F.pad(want_w, (0, info["cin_p"] - info["cin"], 0, info["kw_p"] - info["k"])).reshape(info["cout"], -1)
    errs = {}
    for sp in (0, 9, 6):
        e = engs[sp]
        assert e.products == sp
        out = torch.empty((imgs, info["hout"], info["wout"], info["cout"]), device=dev)
        e.debug_conv(0, ci, x_d, None, out, imgs, 1, torch.empty((1, 2, info["cout"]), device=dev))
        dx = torch.empty((imgs, info["hin"], info["win"], info["cin"]), device=dev)
        if ci > 0:
            e.debug_conv(1, ci, None, dy_d, dx, imgs)
        else:
            dx = _nhwc(xr.grad.float(), info["cin"]).to(dev)        # the stem needs no input gradient
        dw = torch.empty((info["cout"], info["Kw"]), device=dev)
        e.debug_conv(2, ci, x_d, dy_d, dw, imgs)
        rel = lambda got, want: ((got.double() - want).norm() / want.norm()).item()
        errs[sp] = (rel(out.cpu().permute(0, 3, 1, 2), y.detach()), rel(dx.cpu().permute(0, 3, 1, 2), xr.grad),
                    rel(dw.cpu()[:, :want_w.shape[1]], want_w))
    print(f"conv{ci} relative L2 error vs float64 (fwd, dgrad, wgrad): {errs}")
    for sp in (9, 6):
        for k in range(3):
            assert errs[sp][k] <= 1.25 * errs[0][k] + 1e-9, (sp, k, errs)
            assert errs[sp][k] < 2e-6


@pytest.mark.parametrize("ci", list(range(20)))
def test_conv_all_layers_64(eng64, ci):
    e, sd = eng64
    _check_conv(e, sd, ci, imgs=6, groups=2, seed=100 + ci)


@pytest.mark.parametrize("ci", [0, 1, 5, 7, 10, 12, 15, 17, 19])
def test_conv_layers_224(eng224, ci):
    e, sd = eng224
    _check_conv(e, sd, ci, imgs=3, groups=1, seed=200 + ci)


@pytest.mark.parametrize("hw,imgs,groups", [(64, 6, 2), (224, 3, 1), (96, 2, 1)])
@pytest.mark.parametrize("packed", [1, 0])
def test_stem_conv_forms(monkeypatch, hw, imgs, groups, packed):
    """Forward (+ BN partials) and weight gradient of the 7x7 stem against F.conv2d in both K layouts: packed (default;
    zero-framed NHWC3 input, kernel rows of 24 floats, K = 176 -- the gradient slot of the zero tap, columns 21..23 of
    every row, and the 8 pad columns must come out 0) and FM_STEM_PACKED=0 ([7][8][4], K = 224)."""
    from fedmlp_amd.engine import Engine
    monkeypatch.setenv("FM_STEM_PACKED", str(packed))
    e = Engine("Resnet18", 5, hw, hw, 2 * imgs)
    try:
        flat, cnt = spec.init_state("Resnet18", 5, 11)
        e.set_state(flat, cnt)
        info = e.debug_conv_info(0)
        if packed:
            assert info["Kw"] == 176 and info["cin_p"] == 3 and info["kw_p"] == 8
        else:
            assert info["Kw"] == 224 and info["cin_p"] == 4 and info["kw_p"] == 8
        _check_conv(e, spec.flat_to_state_dict("Resnet18", 5, flat, cnt), 0, imgs=imgs, groups=groups, seed=300 + hw)
    finally:
        e.close()


def test_conv_streamk_forced_splits():
    """Stream-K fix-up paths (partial tiles summed by the last arriver) on small shapes: the
    persistent grid is forced to odd block counts in a child process (the grid override is
    read once per process) and every conv layer is re-checked."""
    import os, subprocess, sys
    code = (
        "import sys; sys.path.insert(0, '.');"
        "import tests.test_kernels_gpu as T; from fedmlp_amd.engine import Engine; from fedmlp_amd import spec;"
        "e = Engine('Resnet18', 5, 64, 64, 8); flat, cnt = spec.init_state('Resnet18', 5, 1037);"
        "e.set_state(flat, cnt); sd = spec.flat_to_state_dict('Resnet18', 5, flat, cnt);"
        "[T._check_conv(e, sd, ci, 6, 2, 300 + ci) for ci in range(20)]; e.close(); print('ok')")
    for nb in ("7", "61", "509"):
        env = dict(os.environ, FM_IGEMM_BLOCKS=nb)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, f"FM_IGEMM_BLOCKS={nb}: {r.stdout[-2000:]} {r.stderr[-3000:]}"


def test_state_roundtrip(eng64):
    e, sd = eng64
    flat, cnt = spec.init_state("Resnet18", 5, 99)
    cnt[:] = np.arange(len(cnt))
    e.set_state(flat, cnt)
    f2, c2 = e.get_state()
    np.testing.assert_array_equal(f2, flat)
    np.testing.assert_array_equal(c2, cnt)
    flat0, cnt0 = spec.init_state("Resnet18", 5, 1037)
    e.set_state(flat0, cnt0)
