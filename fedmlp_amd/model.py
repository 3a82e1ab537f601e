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
        keys = [k for k, _, _ in spec.entries(self.model, self.n_classes)]
        if strict and set(keys) != set(sd.keys()):
            missing = sorted(set(keys) - set(sd.keys())); extra = sorted(set(sd.keys()) - set(keys))
            raise RuntimeError(f"load_state_dict: missing {missing[:4]}..., unexpected {extra[:4]}...")
        self.flat, self.counters = spec.state_dict_to_flat(self.model, self.n_classes, sd)
        self._touch()
        return self

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

    def bind(self, in_h, in_w, max_images, device="cuda:0"):
        """Make this net's state the engine's resident state and return the engine."""
        eng = get_engine(self.model, self.n_classes, in_h, in_w, max_images, device)
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
        if eng is not None and getattr(eng, "_owner", None) is self and getattr(eng, "_dirty", False):
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
        self.resident = True

    def bind(self, in_h, in_w, max_images, device="cuda:0"):
        eng = self._engine
        assert (in_h, in_w) == (eng.in_h, eng.in_w) and max_images <= eng.max_images, \
            "resident engine was created for another input size / batch"
        return eng

    def _pull(self):
        self.flat, self.counters = self._engine.get_state()

    def mark_trained(self):
        pass

    def load_state_dict(self, sd, strict=True):
        flat, cnt = spec.state_dict_to_flat(self.model, self.n_classes, sd)
        self._engine.set_state(flat, cnt)
        return self

    def __deepcopy__(self, memo):
        return self          # "deepcopy(netglob)" of a resident net is the resident net


def build_model(args):
    """model/build_model.py:5-10.  Reads args.model, args.n_classes, args.pretrained.
    ImageNet weights (args.pretrained, utils/options.py:26) cannot be downloaded
    offline: initialisation is the deterministic from-scratch policy of
    spec.init_state seeded with args.seed; load real weights with load_state_dict."""
    name = getattr(args, "model", "Resnet18")
    if name not in _MODEL_ALIASES:
        raise ValueError(f"build_model: model {name!r} is not built (available: Resnet18, Efficient_b0)")
    seed = int(getattr(args, "seed", 1037))
    flat, cnt = spec.init_state(_MODEL_ALIASES[name], args.n_classes, seed)
    net = HipNet(name, args.n_classes, flat, cnt)
    net.default_max_images = 4 * int(getattr(args, "batch_size", 32))
    return net
