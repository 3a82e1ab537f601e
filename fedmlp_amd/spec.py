"""state_dict layout of the models the engine trains (host side, no GPU needed).

The reference's model factory is ``model/build_model.py:5-10`` ->
``model/all_models.py:53-54, 117-120`` (torchvision resnet18 + Linear(512, C)).
Checkpoints and FedAvg (``utils/FedAvg.py:7-14``) walk the torchvision
state_dict, so the engine's export/import order is exactly that key order:
``conv1.weight, bn1.{weight,bias,running_mean,running_var,num_batches_tracked},
layer1.0.conv1.weight, ..., fc.weight, fc.bias`` (122 entries for ResNet-18).

The C-ABI moves the state as ONE flat fp32 buffer (every float entry
concatenated in key order, conv weights in OIHW) plus ONE int64 buffer (the
``num_batches_tracked`` counters in key order); see include/fedmlp_hip.h.
"""
from collections import OrderedDict

import numpy as np

FEATURE_DIM = {"Resnet18": 512, "Efficient_b0": 1280}

# EfficientNet-B0 stages of efficientnet-pytorch 0.7.1: (repeats, kernel, stride, expand, in, out)
B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
             (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def b0_blocks():
    """[(kernel, stride, expand, cin, cout)] of the 16 MBConv blocks."""
    out = []
    for r, k, s, e, i, o in B0_STAGES:
        out.append((k, s, e, i, o))
        out += [(k, 1, e, o, o)] * (r - 1)
    return out


def resnet18_entries(n_classes):
    """[(key, shape, dtype)] in torchvision state_dict order."""
    ent = []

    def conv(name, cout, cin, k):
        ent.append((name + ".weight", (cout, cin, k, k), "f32"))

    def bn(name, c):
        ent.append((name + ".weight", (c,), "f32"))
        ent.append((name + ".bias", (c,), "f32"))
        ent.append((name + ".running_mean", (c,), "f32"))
        ent.append((name + ".running_var", (c,), "f32"))
        ent.append((name + ".num_batches_tracked", (), "i64"))

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    cin = 64
    for li, w in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"layer{li}.{b}"
            stride = 2 if (li > 1 and b == 0) else 1
            conv(p + ".conv1", w, cin, 3)
            bn(p + ".bn1", w)
            conv(p + ".conv2", w, w, 3)
            bn(p + ".bn2", w)
            if stride != 1 or cin != w:
                conv(p + ".downsample.0", w, cin, 1)
                bn(p + ".downsample.1", w)
            cin = w
    ent.append(("fc.weight", (n_classes, 512), "f32"))
    ent.append(("fc.bias", (n_classes,), "f32"))
    return ent


def efficientnet_b0_entries(n_classes):
    """[(key, shape, dtype)] in efficientnet-pytorch 0.7.1 state_dict order (the reference builds it at
    model/efficientnet.py:28-33 and swaps `_fc`, model/all_models.py:121-124)."""
    ent = []

    def bn(name, c):
        ent.append((name + ".weight", (c,), "f32"))
        ent.append((name + ".bias", (c,), "f32"))
        ent.append((name + ".running_mean", (c,), "f32"))
        ent.append((name + ".running_var", (c,), "f32"))
        ent.append((name + ".num_batches_tracked", (), "i64"))

    ent.append(("_conv_stem.weight", (32, 3, 3, 3), "f32"))
    bn("_bn0", 32)
    for i, (k, s, e, cin, cout) in enumerate(b0_blocks()):
        p = f"_blocks.{i}"
        ce = cin * e
        if e != 1:
            ent.append((p + "._expand_conv.weight", (ce, cin, 1, 1), "f32"))
            bn(p + "._bn0", ce)
        ent.append((p + "._depthwise_conv.weight", (ce, 1, k, k), "f32"))
        bn(p + "._bn1", ce)
        cs = max(1, int(cin * 0.25))
        ent.append((p + "._se_reduce.weight", (cs, ce, 1, 1), "f32"))
        ent.append((p + "._se_reduce.bias", (cs,), "f32"))
        ent.append((p + "._se_expand.weight", (ce, cs, 1, 1), "f32"))
        ent.append((p + "._se_expand.bias", (ce,), "f32"))
        ent.append((p + "._project_conv.weight", (cout, ce, 1, 1), "f32"))
        bn(p + "._bn2", cout)
    ent.append(("_conv_head.weight", (1280, 320, 1, 1), "f32"))
    bn("_bn1", 1280)
    ent.append(("_fc.weight", (n_classes, 1280), "f32"))
    ent.append(("_fc.bias", (n_classes,), "f32"))
    return ent


def entries(model, n_classes):
    if model == "Resnet18":
        return resnet18_entries(n_classes)
    if model == "Efficient_b0":
        return efficientnet_b0_entries(n_classes)
    raise ValueError(f"unsupported model {model!r} (built: Resnet18, Efficient_b0)")


def sizes(model, n_classes):
    """(number of fp32 elements, number of int64 counters) of the flat state."""
    nf = ni = 0
    for _, shape, dt in entries(model, n_classes):
        n = int(np.prod(shape)) if shape else 1
        if dt == "f32":
            nf += n
        else:
            ni += n
    return nf, ni


def is_trainable(key):
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


def init_state(model, n_classes, seed):
    """Deterministic from-scratch initialisation (numpy RandomState, so both the
    build container and the GPU box regenerate it bit-identically).  Follows
    torchvision's resnet init policy: conv kaiming-normal(fan_out, relu), BN
    gamma=1 beta=0, running stats 0/1, Linear U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
    (--pretrained 1 weights, utils/options.py:26, cannot be fetched offline.)
    Returns (flat_f32, counters_i64)."""
    rs = np.random.RandomState(seed)
    fl, cnt = [], []
    for key, shape, dt in entries(model, n_classes):
        if dt == "i64":
            cnt.append(0)
            continue
        if len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            v = rs.standard_normal(shape).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_out))
        elif key.startswith("fc.") or key.startswith("_fc."):
            bound = 1.0 / np.sqrt(float(FEATURE_DIM[model]))
            v = rs.uniform(-bound, bound, size=shape).astype(np.float32)
        elif key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            v = np.ones(shape, np.float32)
        else:
            v = np.zeros(shape, np.float32)
        fl.append(v.reshape(-1))
    return np.concatenate(fl), np.array(cnt, dtype=np.int64)


def flat_to_state_dict(model, n_classes, flat, counters):
    """Flat buffers -> OrderedDict[str, np.ndarray] with reference key names."""
    out, of, oc = OrderedDict(), 0, 0
    for key, shape, dt in entries(model, n_classes):
        if dt == "i64":
            out[key] = np.array(counters[oc], dtype=np.int64)
            oc += 1
        else:
            n = int(np.prod(shape))
            out[key] = np.asarray(flat[of:of + n], dtype=np.float32).reshape(shape).copy()
            of += n
    return out


def state_dict_to_flat(model, n_classes, sd):
    """Inverse of flat_to_state_dict; accepts numpy arrays or torch tensors.
    A float-valued counter (what FedAvg produces, utils/FedAvg.py:13) is
    truncated like load_state_dict's copy into an int64 buffer does."""
    fl, cnt = [], []
    for key, shape, dt in entries(model, n_classes):
        v = sd[key]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if dt == "i64":
            cnt.append(int(np.trunc(float(np.asarray(v).reshape(-1)[0]))))
        else:
            assert tuple(v.shape) == tuple(shape), (key, v.shape, shape)
            fl.append(v.astype(np.float32).reshape(-1))
    return np.concatenate(fl), np.array(cnt, dtype=np.int64)


def merge_state_dict(model, n_classes, sd, flat, counters):
    """load_state_dict(strict=False): entries of `sd` whose key exists and whose shape matches
    replace the current value; missing keys and shape-mismatched entries (an ImageNet fc
    [1000, D] loaded into a C-class model, model/all_models.py:117-124) keep the current
    value.  Returns (flat, counters, missing_keys, unexpected_keys, mismatched_keys)."""
    flat = np.array(flat, dtype=np.float32, copy=True)
    counters = np.array(counters, dtype=np.int64, copy=True)
    missing, mismatched = [], []
    known = set()
    of = oc = 0
    for key, shape, dt in entries(model, n_classes):
        known.add(key)
        n = int(np.prod(shape)) if shape else 1
        v = sd.get(key) if hasattr(sd, "get") else None
        if v is None:
            missing.append(key)
        else:
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            if dt == "i64":
                counters[oc] = int(np.trunc(float(v.reshape(-1)[0])))
            elif tuple(v.shape) != tuple(shape):
                mismatched.append(key)
            else:
                flat[of:of + n] = v.astype(np.float32).reshape(-1)
        if dt == "i64":
            oc += 1
        else:
            of += n
    unexpected = [k for k in sd.keys() if k not in known]
    return flat, counters, missing, unexpected, mismatched


def classifier_keys(model):
    """The layer modify_last_layer replaces (model/all_models.py:117-124)."""
    return ("fc.weight", "fc.bias") if model == "Resnet18" else ("_fc.weight", "_fc.bias")
