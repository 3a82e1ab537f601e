"""Python handle over the C-ABI engine (include/fedmlp_hip.h).

torch is used for device memory and streams only (tensor.data_ptr() crosses the
ABI); all arithmetic on the path runs in the HIP library.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, spec

MODEL_IDS = {"Resnet18": 0, "Efficient_b0": 1}
PRECISION_IDS = {"fp32": 0, "bf16": 1}
# fm_config.reserved[2]: None = the library default (six products unless FM_MFMA_SPLIT says otherwise), 0 = fp32 matrix pipe,
# 9 = all nine partial products, 6 = exactly six (an explicit request is not subject to the environment override)
PRODUCT_FORMS = {None: 0, 6: 3, 0: 1, 9: 2}


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.int32, torch.bfloat16), \
        (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


class Engine:
    """One per process/GPU. Owns the device-resident model state, optimiser
    moments, teacher snapshot and activation workspaces for `max_images`."""

    def __init__(self, model, n_classes, in_h, in_w, max_images, device=None, precision="fp32", streams=0, products=None):
        """streams: fm_config.reserved[1] -- 0 the engine forks its side stream for the frozen teacher and the weight
        gradients (default, bit-identical to one stream), 1 one stream (per-kernel profiling), 2 teacher only.
        products: fm_config.reserved[2] -- how the fp32 conv GEMMs form their products, fixed for the handle: None = library
        default (six exact bf16 partial products per fp32 product on the bf16 matrix pipe), 0 = fp32 matrix pipe, 6, 9."""
        if not torch.cuda.is_available():
            raise RuntimeError("fedmlp_amd.Engine needs a GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.model, self.n_classes = model, int(n_classes)
        self.in_h, self.in_w, self.max_images = int(in_h), int(in_w), int(max_images)
        if device is None:                       # the rank's own GPU, never a hard-coded cuda:0
            from .launch import default_device
            device = default_device()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.precision = precision
        if precision != "fp32" and model != "Efficient_b0":
            raise ValueError("precision 'bf16' is built for Efficient_b0 only (BASELINE configs[4])")
        # the engine enqueues on torch's CURRENT stream of this device (0 = the null stream when it is the default one),
        # so its kernels are ordered with the torch ops around the calls
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self.streams = int(streams)
        cfg = _lib.FmConfig(MODEL_IDS[model], self.n_classes, self.in_h, self.in_w,
                            self.max_images, (C.c_int32 * 3)(PRECISION_IDS[precision], self.streams,
                                                             PRODUCT_FORMS[products]),
                            C.c_void_p(self.stream) if self.stream else None)
        h = C.c_void_p()
        _lib.check(self.lib.fm_create(C.byref(cfg), C.byref(h)))
        self.h = h
        nf, ni = C.c_int64(), C.c_int64()
        _lib.check(self.lib.fm_state_sizes(self.h, C.byref(nf), C.byref(ni)))
        self.nf, self.ni = nf.value, ni.value
        assert (self.nf, self.ni) == spec.sizes(model, self.n_classes), \
            "engine and fedmlp_amd.spec disagree on the state_dict layout"
        self.feature_dim = spec.FEATURE_DIM[model]
        # what fm_create actually set up (it keeps one stream when the second buffer set does not fit in free memory)
        self.stream_mode = int(self.lib.fm_stream_mode(self.h))
        self.products = int(self.lib.fm_products(self.h))      # 0 fp32 matrix pipe, 6 / 9 bf16 partial products
        self.planes = bool(self.lib.fm_planes_mode(self.h))    # conv GEMMs read bf16 planes written by the producing kernels
        # Efficient_b0: draw drop-connect / dropout multipliers before every train step, like the
        # reference's model does inside net(images) in train mode.  Parity tests switch it off and
        # install their own draws with set_stochastic().
        self.stochastic = model == "Efficient_b0"
        self.stochastic_generator = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.fm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- EfficientNet-B0 training-time randomness ---------------------------------
    def set_stochastic(self, drop_connect=None, dropout=None):
        """Multipliers for the NEXT train steps (kept until replaced; None = no drop).
        drop_connect: cuda fp32 [16, imgs]; dropout: cuda fp32 [imgs, 1280] (include/fedmlp_hip.h)."""
        self._dc = None if drop_connect is None else drop_connect.contiguous().float()
        self._dr = None if dropout is None else dropout.contiguous().float()
        _lib.check(self.lib.fm_set_stochastic(self.h, _ptr(self._dc), _ptr(self._dr)))

    def draw_stochastic(self, imgs, generator=None):
        """One draw with efficientnet-pytorch's formulas (drop_connect_rate 0.2 scaled by
        idx/16, dropout 0.2) on the engine's device, installed for the next train step."""
        if self.model != "Efficient_b0":
            return
        dev = self.device
        u = torch.rand((16, imgs), device=dev, generator=generator)
        keep = 1.0 - 0.2 * torch.arange(16, device=dev, dtype=torch.float32).view(16, 1) / 16.0
        dc = torch.floor(keep + u) / keep
        dr = (torch.rand((imgs, 1280), device=dev, generator=generator) >= 0.2).float() / 0.8
        self.set_stochastic(dc, dr)

    # ---- state -----------------------------------------------------------------
    def set_state(self, flat, counters):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        counters = np.ascontiguousarray(counters, dtype=np.int64)
        assert flat.size == self.nf and counters.size == self.ni
        _lib.check(self.lib.fm_set_state(self.h, flat.ctypes.data_as(C.c_void_p),
                                         counters.ctypes.data_as(C.c_void_p)))

    def get_state(self):
        flat = np.empty(self.nf, np.float32)
        counters = np.empty(self.ni, np.int64)
        _lib.check(self.lib.fm_get_state(self.h, flat.ctypes.data_as(C.c_void_p),
                                         counters.ctypes.data_as(C.c_void_p)))
        return flat, counters

    def state_tensor(self):
        """torch view of the engine-layout device state (for the RCCL all-reduce)."""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.fm_state_device(self.h, C.byref(p), C.byref(n)))
        return _device_view(p.value, n.value, self.device)

    def counters(self, new=None):
        c = np.zeros(self.ni, np.int64) if new is None else np.ascontiguousarray(new, dtype=np.int64)
        _lib.check(self.lib.fm_counters(self.h, c.ctypes.data_as(C.c_void_p), int(new is not None)))
        return c

    def state_scale(self, w):
        _lib.check(self.lib.fm_state_scale(self.h, C.c_float(w)))

    def fedavg_fold(self, states, dict_len, out=None):
        """FedAvg (utils/FedAvg.py:7-14) of K engine-layout device states on this GPU, reference order and roundings.
        states: list of cuda fp32 tensors of state_tensor()'s length; out: destination (default: the engine's own state)."""
        K = len(states)
        assert K == len(dict_len) and K >= 1
        for t in states:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.state_tensor().numel()
        ptrs = (C.c_void_p * K)(*[t.data_ptr() for t in states])
        dst = self.state_tensor() if out is None else out
        self._check_stream()
        _lib.check(self.lib.fm_fedavg_fold(self.h, ptrs, _lib.fvec(dict_len, K), K, _ptr(dst)))
        return dst

    def _check_stream(self):
        """The engine enqueues on the stream that was torch's current one when it was built (fm_config.stream is fixed for
        the handle's life: its workspaces are ordered on that stream).  A call made under another torch.cuda.stream(...)
        would race with the producers of its inputs, so it is refused."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if cur != self.stream:
            raise RuntimeError(f"fedmlp_amd.Engine was built on stream {self.stream:#x} but torch's current stream is "
                               f"{cur:#x}: call the engine under the stream it was created on")

    def teacher_snapshot(self):
        _lib.check(self.lib.fm_teacher_snapshot(self.h))

    def adam_reset(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
        hp = _lib.FmAdam(lr, betas[0], betas[1], eps, weight_decay)
        _lib.check(self.lib.fm_adam_reset(self.h, C.byref(hp)))

    def sync(self):
        _lib.check(self.lib.fm_sync(self.h))

    # ---- forward / steps -----------------------------------------------------------
    def forward_eval(self, x, teacher=False):
        B = x.shape[0]
        feat = torch.empty((B, self.feature_dim), device=self.device, dtype=torch.float32)
        logits = torch.empty((B, self.n_classes), device=self.device, dtype=torch.float32)
        return self.forward_eval_into(x, feat, logits, teacher)

    def forward_eval_into(self, x, feat, logits, teacher=False):
        """net(x) in eval mode into caller-owned [B,D] / [B,C] device buffers (no allocation)."""
        self._check_stream()
        _lib.check(self.lib.fm_forward_eval(self.h, _ptr(x), x.shape[0], int(teacher), _ptr(feat), _ptr(logits)))
        return feat, logits

    # ---- RCCL inside the C-ABI library (utils/FedAvg.py:7-14, 51-93 across ranks) ---------
    def comm_preflight(self):
        """librccl loadable and complete on THIS rank (local; fm_comm_init is the collective step)."""
        _lib.check(self.lib.fm_comm_preflight())

    def comm_unique_id(self):
        buf = (C.c_uint8 * _lib.FM_COMM_ID_BYTES)()
        _lib.check(self.lib.fm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * _lib.FM_COMM_ID_BYTES)(*unique_id)
        _lib.check(self.lib.fm_comm_init(self.h, buf, int(rank), int(world)))

    def comm_destroy(self):
        _lib.check(self.lib.fm_comm_destroy(self.h))

    def comm_size(self):
        return int(self.lib.fm_comm_size(self.h))

    def fedavg_allreduce(self, w):
        """state <- sum_ranks w_rank * state_rank on the engine stream (ncclAllReduce in the library)."""
        _lib.check(self.lib.fm_fedavg_allreduce(self.h, C.c_float(w)))

    def fedavg_tao(self, t, n_i, negative_mask):
        n = self.n_classes
        tt = (C.c_double * n)(*[float(v) for v in t])
        out = (C.c_double * n)()
        _lib.check(self.lib.fm_fedavg_tao(self.h, tt, C.c_double(float(n_i)), _lib.fvec(negative_mask, n), out))
        return np.array(out[:], dtype=np.float64)

    def fedavg_proto(self, proto, n_i, active_mask):
        n, D = self.n_classes, self.feature_dim
        p = np.ascontiguousarray(proto, dtype=np.float32).reshape(2 * n, D)
        out = np.empty((2 * n, D), np.float32)
        _lib.check(self.lib.fm_fedavg_proto(self.h, p.ctypes.data_as(C.c_void_p), C.c_double(float(n_i)),
                                            _lib.fvec(active_mask, n), out.ctypes.data_as(C.c_void_p)))
        return out

    def _draw(self, imgs):
        if self.stochastic:
            self.draw_stochastic(imgs, self.stochastic_generator)

    def step_bce(self, x, y, pos_weight, bs_norm, loss_out):
        self._check_stream()
        self._draw(x.shape[0])
        _lib.check(self.lib.fm_step_bce(self.h, _ptr(x), _ptr(y), x.shape[0],
                                        _lib.fvec(pos_weight, self.n_classes), int(bs_norm),
                                        _ptr(loss_out)))

    def step_stage1(self, x1, x2, y, active_mask, annotation_num, bs_norm, loss_out):
        self._check_stream()
        self._draw(2 * x1.shape[0])
        _lib.check(self.lib.fm_step_stage1(self.h, _ptr(x1), _ptr(x2), _ptr(y), x1.shape[0],
                                           _lib.fvec(active_mask, self.n_classes),
                                           int(annotation_num), int(bs_norm), _ptr(loss_out)))

    def step_stage2(self, x, y, distill, loss_out):
        self._check_stream()
        self._draw(x.shape[0])
        _lib.check(self.lib.fm_step_stage2(self.h, _ptr(x), _ptr(y), _ptr(distill), x.shape[0],
                                           _ptr(loss_out)))

    def step_fixmatch(self, xw, xs, y, pos_weight, pos_weight_unk, active_mask, annotation_num,
                      bs_norm, loss_out):
        self._check_stream()
        n = self.n_classes
        self._draw(2 * xw.shape[0])
        _lib.check(self.lib.fm_step_fixmatch(
            self.h, _ptr(xw), _ptr(xs), _ptr(y), xw.shape[0], _lib.fvec(pos_weight, n),
            _lib.fvec(pos_weight_unk, n), _lib.fvec(active_mask, n), int(annotation_num),
            int(bs_norm), _ptr(loss_out)))

    # ---- prototypes / tagging ------------------------------------------------------
    # ---- generic split step (rank-4 baselines: loss head computed by the host mirror) -----------
    def forward_train(self, x1, x2=None):
        self._check_stream()
        views = 1 if x2 is None else 2
        B = x1.shape[0]
        self._draw(views * B)
        feat = torch.empty((views * B, self.feature_dim), device=self.device, dtype=torch.float32)
        logits = torch.empty((views * B, self.n_classes), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.fm_forward_train(self.h, _ptr(x1), _ptr(x2), B, _ptr(feat), _ptr(logits)))
        return feat, logits

    def backward_step(self, dlogits):
        self._check_stream()
        _lib.check(self.lib.fm_backward_step(self.h, _ptr(dlogits.contiguous().float())))

    def teacher_axpby(self, w_teacher, w_student):
        _lib.check(self.lib.fm_teacher_axpby(self.h, C.c_float(w_teacher), C.c_float(w_student)))

    def teacher_swap(self):
        _lib.check(self.lib.fm_teacher_swap(self.h))

    def proto_reset(self):
        _lib.check(self.lib.fm_proto_reset(self.h))

    def proto_accumulate(self, feat, logits, labels, active_mask, negative_mask, L, U):
        n = self.n_classes
        _lib.check(self.lib.fm_proto_accumulate(self.h, _ptr(feat), _ptr(logits), _ptr(labels),
                                                feat.shape[0], _lib.fvec(active_mask, n),
                                                _lib.fvec(negative_mask, n), C.c_float(L),
                                                C.c_float(U)))

    def proto_finalize(self, zero_guard, n_local, active_mask):
        proto = np.empty((2 * self.n_classes, self.feature_dim), np.float32)
        t = np.empty(self.n_classes, np.float64)
        _lib.check(self.lib.fm_proto_finalize(self.h, int(zero_guard), int(n_local),
                                              _lib.fvec(active_mask, self.n_classes),
                                              proto.ctypes.data_as(C.c_void_p),
                                              t.ctypes.data_as(C.c_void_p)))
        return t, proto

    def cos_tag(self, feat, proto, classes):
        N = feat.shape[0]
        sim = torch.empty((len(classes), N), device=self.device, dtype=torch.float32)
        if len(classes) and N:
            cls = (C.c_int32 * len(classes))(*[int(c) for c in classes])
            _lib.check(self.lib.fm_cos_tag(self.h, _ptr(feat), N, _ptr(proto), cls, len(classes),
                                           _ptr(sim)))
        return sim

    def select_topk(self, sim_row, clean_thr, noise_thr):
        N = sim_row.shape[0]
        cap = max(N, 1)
        top, bot = (C.c_int32 * cap)(), (C.c_int32 * cap)()
        nt, nb = C.c_int32(), C.c_int32()
        _lib.check(self.lib.fm_select_topk(self.h, _ptr(sim_row), N, float(clean_thr),
                                           float(noise_thr), cap, top, C.byref(nt), bot,
                                           C.byref(nb)))
        return list(top[:nt.value]), list(bot[:nb.value])

    def select_topk_rows(self, sims, pools, clean_thr, noise_thr):
        """Every class of a round at once: sims [n_cls, N] (cos_tag's output), pools[k] = the positions of class k's pool inside
        its similarity row, in pool order (None = every row).  -> [(top, bot)] per class, as positions INSIDE the pool.  One
        launch pair and one device-to-host read (fm_select_topk_rows)."""
        n_cls, N = int(sims.shape[0]), int(sims.shape[1])
        if n_cls == 0:
            return []
        sizes = [N if p is None else len(p) for p in pools]
        whole = all(p is None for p in pools)
        stride = max(max(sizes), 1)
        rows = None
        if not whole:
            flat = np.zeros((n_cls, stride), dtype=np.int32)
            for k, p in enumerate(pools):
                flat[k, :sizes[k]] = np.arange(N, dtype=np.int32) if p is None else np.asarray(p, dtype=np.int32)
            rows = flat.ctypes.data_as(C.POINTER(C.c_int32))
            self._keep = flat
        cap = int(max(clean_thr, noise_thr, 0.0) * max(sizes)) + 1
        pn = (C.c_int32 * n_cls)(*sizes)
        top, bot = (C.c_int32 * (n_cls * cap))(), (C.c_int32 * (n_cls * cap))()
        nt, nb = (C.c_int32 * n_cls)(), (C.c_int32 * n_cls)()
        _lib.check(self.lib.fm_select_topk_rows(self.h, _ptr(sims), N, n_cls, rows, pn, stride, float(clean_thr),
                                                float(noise_thr), cap, top, nt, bot, nb))
        return [(list(top[k * cap:k * cap + nt[k]]), list(bot[k * cap:k * cap + nb[k]])) for k in range(n_cls)]

    # ---- input pipeline ---------------------------------------------------------------
    def augment(self, cache_u8, idx, params, mean, std):
        """uint8 cache [N,3,H,W] + sample indices [B] (int32) + fixed-point affine/flip records [B,8] (int32,
        fedmlp_amd.augment.fixed_point_params) -> fp32 NCHW batch."""
        B = idx.shape[0]
        out = torch.empty((B, 3, self.in_h, self.in_w), device=self.device, dtype=torch.float32)
        assert cache_u8.is_cuda and cache_u8.dtype == torch.uint8 and cache_u8.is_contiguous()
        assert params.dtype == torch.int32 and params.shape == (B, 8)
        _lib.check(self.lib.fm_augment(self.h, C.c_void_p(cache_u8.data_ptr()), _ptr(idx), _ptr(params), B,
                                       _lib.fvec(mean, 3), _lib.fvec(std, 3), _ptr(out)))
        return out

    # ---- measurement ------------------------------------------------------------------
    def profile_enable(self, on):
        _lib.check(self.lib.fm_profile_enable(self.h, int(on)))

    def profile_read(self, family):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib.fm_profile_read(self.h, family, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def profile_ops(self, enable, read=True):
        """[(label, calls, total_ms)] of the ops timed since the last read (EfficientNet-B0 graph)."""
        buf = C.create_string_buffer(1 << 16) if read else None
        _lib.check(self.lib.fm_profile_ops(self.h, int(enable), buf, (1 << 16) if read else 0))
        rows = []
        if read:
            for line in buf.value.decode().splitlines():
                lab, n, ms = line.split("\t")
                rows.append((lab, int(n), float(ms)))
        return rows

    # ---- kernel-level test hooks ----------------------------------------------------------
    def debug_conv_info(self, conv):
        info = (C.c_int32 * 16)()
        _lib.check(self.lib.fm_debug_conv_info(self.h, conv, info))
        keys = ("cin", "cout", "k", "stride", "pad", "hin", "win", "hout", "wout", "cin_p", "Kw", "kw_p", "cout_p")
        return dict(zip(keys, list(info)))

    def debug_stem_masks(self, imgs, groups):
        """(relu mask [imgs, H/2, W/2, 64] bool, max-pool argmax code [imgs, H/4, W/4, 64] uint8) of the last train-mode forward's
        stem (fm_debug_stem_masks)"""
        H2, W2 = self.in_h // 2, self.in_w // 2
        bits = np.empty((imgs, H2, W2, 8), np.uint8)
        code = np.empty((imgs, H2 // 2, W2 // 2, 64), np.uint8)
        _lib.check(self.lib.fm_debug_stem_masks(self.h, imgs, groups, bits.ctypes.data_as(C.c_void_p),
                                                code.ctypes.data_as(C.c_void_p)))
        return np.unpackbits(bits, axis=-1, bitorder="little").astype(bool), code

    def debug_get_grads(self):
        flat = np.empty(self.nf, np.float32)
        _lib.check(self.lib.fm_debug_get_grads(self.h, flat.ctypes.data_as(C.c_void_p)))
        return flat

    def debug_pw(self, op, conv, x, dy, out, imgs, groups=1, psc=None, psh=None, gate=None, stats=None):
        _lib.check(self.lib.fm_debug_pw(self.h, op, conv, _ptr(x), _ptr(dy), _ptr(out), imgs, groups, _ptr(psc),
                                        _ptr(psh), _ptr(gate), _ptr(stats)))

    def debug_proj_bwd(self, conv, phase, dyp, yd, bn, gate, ds, imgs, groups, out, pool5=None):
        _lib.check(self.lib.fm_debug_proj_bwd(self.h, conv, phase, _ptr(dyp), _ptr(yd), _ptr(bn), _ptr(gate), _ptr(ds), imgs,
                                              groups, _ptr(out), _ptr(pool5)))

    def debug_exp_bwd(self, conv, da, ye, x, res, bn, imgs, groups, dx, dw):
        _lib.check(self.lib.fm_debug_exp_bwd(self.h, conv, _ptr(da), _ptr(ye), _ptr(x), _ptr(res), _ptr(bn), imgs, groups,
                                             _ptr(dx), _ptr(dw)))

    def debug_activation(self, kind, block, imgs):
        """post-ReLU activation kept by the last train-mode forward, as an NCHW numpy array"""
        dims = (C.c_int32 * 4)()
        _lib.check(self.lib.fm_debug_activation(self.h, kind, block, imgs, None, dims))
        out = np.empty(tuple(dims), np.float32)
        _lib.check(self.lib.fm_debug_activation(self.h, kind, block, imgs, out.ctypes.data_as(C.c_void_p), dims))
        return np.ascontiguousarray(out.transpose(0, 3, 1, 2))

    def debug_num_convs(self):
        return self.lib.fm_debug_num_convs(self.h)

    def debug_conv(self, op, conv, x, dy, out, imgs, groups=1, stats=None):
        _lib.check(self.lib.fm_debug_conv(self.h, op, conv, _ptr(x), _ptr(dy), _ptr(out), imgs, groups,
                                          _ptr(stats)))


class _CudaArrayView:
    """Minimal __cuda_array_interface__ carrier so torch can alias engine memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False),
                                         "version": 2, "strides": None}


def _device_view(ptr, n, device):
    return torch.as_tensor(_CudaArrayView(ptr, n), device=device)


_ENGINES = {}


def get_engine(model, n_classes, in_h, in_w, max_images, device=None, precision="fp32"):
    """Process-wide engine cache (engines own GBs of workspace; the reference's
    cheap deepcopy(net) objects map onto ONE engine whose state is swapped)."""
    if device is None:
        from .launch import default_device
        device = default_device()
    key = (model, int(n_classes), int(in_h), int(in_w), str(device), precision)
    e = _ENGINES.get(key)
    if e is None or e.max_images < max_images:
        if e is not None:
            # a larger workspace replaces the engine: first pull the resident net's device-only
            # trained state to its host copy (and unbind it), or it would be lost with the handle
            owner = getattr(e, "_owner", None)
            if owner is not None:
                owner._pull()
                owner._engine = None
            e._owner = None
            e.close()
        e = Engine(model, n_classes, in_h, in_w, max_images, device, precision)
        _ENGINES[key] = e
    return e


def release_engines():
    """Close every cached engine (pulling a resident net's trained state to its host copy first).  The next
    get_engine() builds a fresh one -- which is also when the library's per-engine environment switches are read."""
    for key, e in list(_ENGINES.items()):
        owner = getattr(e, "_owner", None)
        if owner is not None:
            owner._pull()
            owner._engine = None
        e._owner = None
        e.close()
        del _ENGINES[key]
