#!/usr/bin/env python3
"""ResNet-18 stem forms side by side on the GPU: the [64][7][8][4] padded form (FM_STEM_PACKED=0) and the packed
7 x 24 form (default), same state and batch.  Prints the first-step loss and the per-tensor gradient
difference of the two forms, the same for each form run twice (measured: exactly 0), and the weight difference after
`steps` Adam steps.  Measured at a random init, 64x64 bs 32: losses agree to 2e-7, gradients differ by a median 1.5e-3
of the tensor's norm -- the size of the engine-vs-reference difference at the benchmarked size
(tests/test_golden_r2_gpu.py: 1.1e-3): a random-init ResNet-18's gradient amplifies fp32 rounding of the first conv
by ~1e4.  usage: stem_mode_diff.py [hw] [bs] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedmlp_amd import spec                                    # noqa: E402
from fedmlp_amd.engine import Engine                           # noqa: E402
from fedmlp_amd.model import build_model                       # noqa: E402
from tests.helpers import make_args                            # noqa: E402
from tests.synth import synth_arrays                           # noqa: E402

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
C = 5
targets, x1, x2 = synth_arrays(bs * 4, C, hw, 7, True)
args = make_args(n_classes=C, n_clients=1, batch_size=bs, seed=3, pretrained=0)
net = build_model(args)
flat, cnt = spec.state_dict_to_flat("Resnet18", C, net.state_dict())


def run(padded):
    os.environ["FM_STEM_PACKED"] = "0" if padded else "1"
    eng = Engine("Resnet18", C, hw, hw, 2 * bs)
    eng.set_state(flat, cnt)
    eng.teacher_snapshot()
    eng.adam_reset(args.base_lr)
    y = targets.copy()
    y[:, 1:] = 0.0
    lo = torch.zeros(steps, device=eng.device)
    out = {}
    for k in range(steps):
        sl = slice((k % 4) * bs, (k % 4 + 1) * bs)
        eng.step_stage1(torch.from_numpy(x1[sl]).to(eng.device), torch.from_numpy(x2[sl]).to(eng.device),
                        torch.from_numpy(y[sl]).to(eng.device), [1.0, 0, 0, 0, 0], 1, bs, lo[k:k + 1])
        if k == 0:
            out["grads"] = spec.flat_to_state_dict("Resnet18", C, eng.debug_get_grads(), np.zeros(eng.ni, np.int64))
    out["loss"] = lo.cpu().numpy().astype(np.float64)
    f2, _ = eng.get_state()
    out["sd"] = spec.flat_to_state_dict("Resnet18", C, f2, np.zeros(eng.ni, np.int64))
    eng.close()
    return out


def diff(a, b, tag):
    print(f"== {tag}: loss[0] {a['loss'][0]:.9f} vs {b['loss'][0]:.9f}; loss[-1] {a['loss'][-1]:.9f} vs {b['loss'][-1]:.9f}")
    worst = []
    for k, ga in a["grads"].items():
        gb = b["grads"][k]
        if ga.dtype.kind != "f":
            continue
        worst.append((float(np.linalg.norm(ga.astype(np.float64) - gb)) / (float(np.linalg.norm(ga)) + 1e-30), k))
    worst.sort(reverse=True)
    print("   first-step grads, |a-b|/|a|: worst", [(f"{e:.2e}", k) for e, k in worst[:4]], "median",
          f"{np.median([e for e, _ in worst]):.2e}")
    ws = []
    for k, wa in a["sd"].items():
        if wa.dtype.kind != "f":
            continue
        ws.append((float(np.linalg.norm(wa.astype(np.float64) - b["sd"][k])) / (float(np.linalg.norm(wa)) + 1e-30), k))
    ws.sort(reverse=True)
    print(f"   weights after {steps} steps: worst", [(f"{e:.2e}", k) for e, k in ws[:3]], "median",
          f"{np.median([e for e, _ in ws]):.2e}")


pad, pk1, pk2, pad2 = run(True), run(False), run(False), run(True)
diff(pad, pad2, "padded vs padded (run to run)")
diff(pk1, pk2, "packed vs packed (run to run)")
diff(pad, pk1, "padded vs packed")
