"""build_model(args) drop-in (reference: model/build_model.py:5-10 ->
model/all_models.py:29-130) returning a model object with the nn.Module surface
the reference's driver touches (main.py:73-77, 181-184, 218-222, 361-366):
``net(x) -> (feature, logits)``, ``train()/eval()``, ``state_dict()/
load_state_dict()`` with torchvision key names, ``to()/cuda()/cpu()``,
``copy.deepcopy``.

A HipNet is a light state container (host copy of the flat state).  The heavy
part -- device weights, optimiser moments, activation workspaces -- lives in the
process-wide HIP engine (fedmlp_amd.engine.get_engine); a net is made resident
on it on demand, so the reference's many ``deepcopy(netglob)`` objects cost a
45 MB host copy each, not a GPU workspace each.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import spec
from .engine import get_engine

_MODEL_ALIASES = {"Resnet18": "Resnet18", "resnet18": "Resnet18", "Efficient_b0": "Efficient_b0"}

# file names torchvision 0.13 / efficientnet-pytorch 0.7.1 download for `pretrained=True`
PRETRAINED_FILES = {"Resnet18": "resnet18-f37072fd.pth", "Efficient_b0": "efficientnet-b0-355c32eb.pth"}


class LoadResult(tuple):
    """(missing_keys, unexpected_keys, mismatched_keys) of a load_state_dict call."""
    def __new__(cls, missing, unexpected, mismatched):
        return super().__new__(cls, (list(missing), list(unexpected), list(mismatched)))
    missing_keys = property(lambda self: self[0])
    unexpected_keys = property(lambda self: self[1])
    mismatched_keys = property(lambda self: self[2])


class HipNet:
    def __init__(self, model, n_classes, flat, counters):
        self.model = _MODEL_ALIASES[model]
        self.n_classes = int(n_classes)
        self.flat = np.ascontiguousarray(flat, dtype=np.float32)
        self.counters = np.ascontiguousarray(counters, dtype=np.int64)
        self.training = True
        self._version = 0
        self._engine = None            # engine on which (self, _version) is resident
        self.default_max_images = 128
        self.precision = "fp32"        # activation storage of the engine this net binds to

    # ---- nn.Module-like surface ----------------------------------------------------------
    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def cpu(self):
        return self

    def state_dict(self):
        self._pull()
        sd = spec.flat_to_state_dict(self.model, self.n_classes, self.flat, self.counters)
        return OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items())

    def load_state_dict(self, sd, strict=True):
        """nn.Module.load_state_dict semantics.  strict=False keeps the current value of every
        missing or shape-mismatched entry (so an ImageNet checkpoint with fc [1000, D] loads into a
        C-class net and the classifier keeps its fresh init, like get_model + modify_last_layer,
        model/all_models.py:99-130) and returns (missing, unexpected, mismatched) key lists."""
        keys = [k for k, _, _ in spec.entries(self.model, self.n_classes)]
        if strict:
            if set(keys) != set(sd.keys()):
                missing = sorted(set(keys) - set(sd.keys())); extra = sorted(set(sd.keys()) - set(keys))
                raise RuntimeError(f"load_state_dict: missing {missing[:4]}..., unexpected {extra[:4]}...")
            self.flat, self.counters = spec.state_dict_to_flat(self.model, self.n_classes, sd)
            self._touch()
            return LoadResult([], [], [])
        self._pull()
        self.flat, self.counters, missing, unexpected, mismatched = spec.merge_state_dict(
            self.model, self.n_classes, sd, self.flat, self.counters)
        self._touch()
        return LoadResult(missing, unexpected, mismatched)

    def parameters(self):
        """Trainable tensors in state_dict order (views of the host copy)."""
        self._pull()
        off = 0
        for key, shape, dt in spec.entries(self.model, self.n_classes):
            if dt != "f32":
                continue
            n = int(np.prod(shape))
            if spec.is_trainable(key):
                yield torch.from_numpy(self.flat[off:off + n].reshape(shape))
            off += n

    def __deepcopy__(self, memo):
        self._pull()
        c = HipNet(self.model, self.n_classes, self.flat.copy(), self.counters.copy())
        c.training = self.training
        c.default_max_images = self.default_max_images
        c.precision = self.precision
        return c

    def __call__(self, x):
        """Eval-mode forward on the HIP engine -> (feature[B,D], logits[B,C]) CUDA tensors.
        (utils/local_training.py:983, 1030, 1227; utils/evaluations.py:25.)  The
        train-mode forward is not exposed as an autograd graph: the fused training
        steps live behind LocalUpdate.train*/Engine.step_*."""
        if self.training:
            raise NotImplementedError(
                "HipNet(x) runs the eval-mode forward only; call net.eval() first, or use "
                "fedmlp_amd.local_training.LocalUpdate / Engine.step_* for training steps")
        x = torch.as_tensor(x, dtype=torch.float32)
        eng = self.bind(x.shape[2], x.shape[3], max(self.default_max_images, x.shape[0]))
        return eng.forward_eval(x.to(eng.device).contiguous())

    forward = __call__

    # ---- residency -------------------------------------------------------------------------
    def _touch(self):
        self._version += 1
        self._engine = None

    def bind(self, in_h, in_w, max_images, device=None):
        """Make this net's state the engine's resident state and return the engine
        (device None = this rank's GPU, fedmlp_amd.launch.default_device)."""
        eng = get_engine(self.model, self.n_classes, in_h, in_w, max_images, device, self.precision)
        if getattr(eng, "_owner", None) is not self or getattr(eng, "_owner_version", -1) != self._version \
                or self._engine is not eng:
            prev = getattr(eng, "_owner", None)
            if prev is not None and prev is not self:
                prev._pull()               # do not lose another net's trained, device-only state
            self._pull()
            eng.set_state(self.flat, self.counters)
            eng._owner, eng._owner_version, eng._dirty = self, self._version, False
            self._engine = eng
        return eng

    def _pull(self):
        """Refresh the host copy if the engine trained this net since it was bound."""
        eng = self._engine
        if eng is not None and eng.h and getattr(eng, "_owner", None) is self and getattr(eng, "_dirty", False):
            self.flat, self.counters = eng.get_state()
            eng._dirty = False

    def mark_trained(self):
        if self._engine is not None:
            self._engine._dirty = True


class ResidentNet(HipNet):
    """A net whose state already lives in an engine (one client per GPU: the model never
    leaves HBM between rounds; FedAvg is an in-place RCCL all-reduce of the engine state).
    bind() hands back that engine without any upload; state_dict() still works (D2H copy)."""

    def __init__(self, engine):
        self.model, self.n_classes = engine.model, engine.n_classes
        self.training = True
        self._version = 0
        self._engine = engine
        self.default_max_images = engine.max_images
        self.precision = engine.precision
        self.resident = True

    def bind(self, in_h, in_w, max_images, device=None):
        eng = self._engine
        assert (in_h, in_w) == (eng.in_h, eng.in_w) and max_images <= eng.max_images, \
            "resident engine was created for another input size / batch"
        return eng

    def _pull(self):
        self.flat, self.counters = self._engine.get_state()

    def mark_trained(self):
        pass

    def load_state_dict(self, sd, strict=True):
        if strict:
            flat, cnt = spec.state_dict_to_flat(self.model, self.n_classes, sd)
            res = LoadResult([], [], [])
        else:
            f0, c0 = self._engine.get_state()
            flat, cnt, *lists = spec.merge_state_dict(self.model, self.n_classes, sd, f0, c0)
            res = LoadResult(*lists)
        self._engine.set_state(flat, cnt)
        return res

    def __deepcopy__(self, memo):
        return self          # "deepcopy(netglob)" of a resident net is the resident net


def find_pretrained(name, args=None):
    """Where an ImageNet checkpoint for `name` may already sit on this machine: args.pretrained_path,
    $FEDMLP_PRETRAINED_DIR, then torch hub's checkpoint cache (what `pretrained=True` populates in
    the reference, model/all_models.py:53-54, 73-75).  None if nothing is there (no network here)."""
    import os
    cands = []
    p = getattr(args, "pretrained_path", None) if args is not None else None
    if p:
        cands.append(p if not os.path.isdir(p) else os.path.join(p, PRETRAINED_FILES[name]))
    if os.environ.get("FEDMLP_PRETRAINED_DIR"):
        cands.append(os.path.join(os.environ["FEDMLP_PRETRAINED_DIR"], PRETRAINED_FILES[name]))
    hub = os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch"))
    cands.append(os.path.join(hub, "hub", "checkpoints", PRETRAINED_FILES[name]))
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def build_model(args):
    """model/build_model.py:5-10: get_model(args.model, args.pretrained) + modify_last_layer.
    Reads args.model, args.n_classes, args.pretrained (default 1 in utils/options.py:26),
    optional args.pretrained_path / args.precision.
    pretrained: the ImageNet state_dict is loaded from a local file (find_pretrained) with
    strict=False, so every backbone entry is taken and the 1000-class classifier is dropped in
    favour of a fresh Linear(D, n_classes) -- exactly what modify_last_layer leaves.  Without a
    file (this image has no network) a warning is raised and training starts from the
    deterministic from-scratch init of spec.init_state seeded with args.seed."""
    import warnings
    name = getattr(args, "model", "Resnet18")
    if name not in _MODEL_ALIASES:
        raise ValueError(f"build_model: model {name!r} is not built (available: Resnet18, Efficient_b0)")
    seed = int(getattr(args, "seed", 1037))
    flat, cnt = spec.init_state(_MODEL_ALIASES[name], args.n_classes, seed)
    net = HipNet(name, args.n_classes, flat, cnt)
    net.default_max_images = 4 * int(getattr(args, "batch_size", 32))
    net.precision = getattr(args, "precision", "fp32")
    if int(getattr(args, "pretrained", 0)):
        path = find_pretrained(net.model, args)
        if path is None:
            warnings.warn(
                f"build_model: args.pretrained={args.pretrained} but no ImageNet checkpoint "
                f"({PRETRAINED_FILES[net.model]}) was found (args.pretrained_path, $FEDMLP_PRETRAINED_DIR, "
                "torch hub cache) and this machine cannot download one: training starts from the "
                "from-scratch init. Pass --pretrained 0 to silence this.", RuntimeWarning, stacklevel=2)
        else:
            import torch as _t
            sd = _t.load(path, map_location="cpu")
            res = net.load_state_dict(sd, strict=False)
            # torchvision's / efficientnet-pytorch's published ImageNet files carry no `num_batches_tracked` entries: torch's
            # BatchNorm leaves a missing counter at 0 (the reference loads such files), and so does merge_state_dict
            bad = [k for k in res.missing_keys + res.mismatched_keys
                   if k not in spec.classifier_keys(net.model) and not k.endswith(".num_batches_tracked")]
            if bad:
                raise RuntimeError(f"build_model: {path} does not match {net.model}: {bad[:6]}")
    return net
