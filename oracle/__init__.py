"""CPU oracle for the FedMLP per-client training hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``fedmlp_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may. It restates, with plain ``torch`` CPU fp32 ops, the
arithmetic of the reference path (``utils/local_training.py``,
``utils/FedAvg.py``, ``utils/FedNoRo.py``, ``utils/utils.py``) so that the HIP
engine can be checked against it on a box where ``/root/reference`` does not
exist.

Parity pinning: the trainer-level functions (losses, DatasetSplit masking,
prototype pass, cosine tagging, top-k selection, FedAvg*) are pinned against
golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.json|npz``).
The MODEL arithmetic (ResNet-18 ``forward -> (feature, logits)``) lives in
un-vendored, locally patched third-party packages (torchvision==0.13.1,
``requirements.txt:102``) and is therefore "parity unpinned" at the model
boundary: the yardstick is this package's own restatement of the
torchvision-0.13 topology, driven through the reference trainer for the
trajectory goldens.
"""
