"""ctypes binding of ``libcraft_hip.so`` (C ABI in ``include/craft_hip.h``).

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``) and the current HIP
stream.  There is NO fallback: if the shared library is missing or a call fails, an exception is
raised — the product path never silently runs on PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_void_p, c_char_p

import torch

ABI_VERSION = 4          # == CRAFT_HIP_ABI_VERSION (include/craft_hip.h): bumped whenever an existing entry point changes its signature or goes away
PREC_F32, PREC_BF16, PREC_F16, PREC_F16X3 = 0, 1, 2, 3
ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
W_PACKED = 0x100
W1X1_PACKED = 0x800           # == CRAFT_W1X1_PACKED: craft_motion_encoder's convc1 weights from ops.pack_linear_weight
CONV_W16 = 0x400              # == CRAFT_CONV_W16: hi plane of the packed weights only (input-gradient convolutions, role wgx)


def WGRAD_X_PREC(p: int) -> int:
    """== CRAFT_WGRAD_X_PREC(p): or-ed into craft_wgrad_pk's prec when the X packs' mode differs from dY's"""
    return (p + 1) << 8

PYR_TILED = 0x200             # == CRAFT_PYR_TILED: levels 0 / 1 of a correlation pyramid in 8x16 / 4x8 tiles
P_TILED = 0x400               # == CRAFT_P_TILED: attention probabilities in 32-query x 64-key tiles (probs_fused -> attn_apply)
PV_ROWS_SHIFT = 20            # CRAFT_PV_ROWS(r) = r << 20, or-ed into craft_attn_apply's prec
FRAG_ACC_ORDER = 0x10000      # == CRAFT_FRAG_ACC_ORDER
STATS_REPLICAS = 64   # CRAFT_STATS_REPLICAS
PREC_NAMES = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "fp16": PREC_F16, "f16": PREC_F16, "f16x3": PREC_F16X3,
              "fp16x3": PREC_F16X3}
PROB_DTYPE = {PREC_F32: torch.float32, PREC_BF16: torch.bfloat16, PREC_F16: torch.float16}

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcraft_hip.so")
_lib = None

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> argtypes; every function returns int.  Kept in sync with include/craft_hip.h
# (tests/test_abi.py checks that each declared symbol is exported).
_SIGS = {
    "craft_tokens": [P, I, I, I, I, I, I, L, I, I, P, L, P],
    "craft_tokens_to_nchw": [P, L, I, I, I, P, P],
    "craft_linear": [P, L, P, P, P, L, L, I, I, I, P],
    "craft_linear_t": [P, L, P, P, L, I, I, I, I, I, I, I, P],
    "craft_score_max": [P, L, P, L, I, I, I, I, I, F, P, I, P],
    "craft_corr_build": [P, L, P, L, I, I, I, I, I, F, P, I, F, F, P, P, P, P, I, P],
    "craft_corr_build_pyramid": [P, L, P, L, I, I, I, I, I, F, P, I, F, F, P, P, P, P, P, P, P, I, P],
    "craft_corr_finish": [P, P, P, P, P, P, I, I, I, I, P],
    "craft_corr_lookup": [P, P, P, P, I, P, P, I, I, I, I, P, L, I, I, P],
    "craft_attn_probs": [P, L, P, L, I, I, I, I, I, F, P, I, F, I, P, P, P, F, P, L, P, I, I, P],
    "craft_attn_apply": [P, L, P, P, I, I, I, I, P, I, P],
    "craft_attn_probs_fused": [P, L, P, L, I, I, I, I, I, F, P, I, F, I, P, P, L, P, P, I, I, P],
    "craft_flash_attention": [P, L, P, L, P, L, I, I, I, I, I, I, F, P, I, F, I, P, P, P, I, I, P],
    "craft_forward_interpolate": [P, I, I, I, P, P],
    "craft_mode_pool_ln": [P, P, L, P, P, I, I, I, I, P, L, P],
    "craft_gma_residual": [P, L, P, P, I, I, I, P, L, P],
    "craft_motion_encoder": [P, L, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, L, P, I, P, P, P],
    "craft_sepconv_gru": [P, L, I, P, P, P, P, P, P, P, P, I, I, I, P, I, P],
    "craft_sepconv_gru_context": [P, L, I, P, P, P, P, P, P, P, P, I, I, I, P, I, P],
    "craft_sepconv_gru_step": [P, L, I, I, P, P, P, P, P, I, I, I, P, I, P],
    "craft_flow_head": [P, L, P, P, P, P, I, I, I, P, P, P, P, P, I, P],
    "craft_mask_head": [P, L, P, P, P, P, I, I, I, P, P, I, P],
    "craft_pack_weights": [P, I, I, I, P, P],
    "craft_conv2d_nhwc": [P, L, I, P, P, I, I, I, I, P, L, I, I, I, I, P],
    "craft_conv2d_nhwc_res": [P, L, I, P, P, I, I, I, I, P, L, P, L, I, I, I, I, P],
    "craft_conv2d_nhwc2_mask": [P, L, I, P, L, I, P, P, P, L, I, I, I, P, L, P, L, I, I, I, I, P],
    "craft_conv2d_nhwc_ex": [P, L, I, I, I, P, P, P, I, I, I, I, I, P, L, I, I, I, P, I, P],
    "craft_stem_conv7x7": [P, P, P, I, I, I, I, P, P, P],
    "craft_stem_conv7x7_mfma": [P, P, P, I, I, I, I, P, P, I, P],
    "craft_stem_conv7x7_mfma_pair": [P, I, P, P, P, I, I, I, I, P, P, I, P],
    "craft_stats_finalize": [P, L, c_double, F, P, P],
    "craft_residual_relu": [P, L, P, P, L, P, I, I, I, I, P, L, P],
    "craft_flow_metrics": [P, P, P, I, I, I, F, F, F, P, P],
    "craft_flow_l1_loss": [P, P, P, I, I, I, F, F, P, P, P],
    "craft_sumsq": [P, L, P, P],
    "craft_pack_conv_weights_batch": [P, P, I, I, P],
    "craft_multi_copy": [P, P, P, P, I, P, P],
    "craft_adamw_step": [P, P, P, P, L, F, F, F, F, F, I, F, P, F, P],
    "craft_loss_scale_update": [P, P, F, F, F, F, F, F, I, P],
    "craft_adamw_step_dyn": [P, P, P, P, L, F, F, F, F, F, P, P],
    "craft_convex_upsample": [P, P, I, I, I, P, P],
    # ---- training
    "craft_gemm": [P, L, L, L, L, P, L, L, L, L, P, L, L, L, I, I, I, I, I, F, I, I, I, P],
    "craft_conv2d_wgrad": [P, L, I, P, L, I, I, I, I, I, I, P, P, P, L, I, P],
    "craft_pack_operand": [P, L, I, L, I, I, I, I, I, L, L, I, P, I, I, P, I, P],
    "craft_gemm_pk": [P, P, P, P, P, L, L, L, I, I, I, I, I, F, I, P],
    "craft_pack_conv_weights": [P, I, P, I, I, I, I, I, I, I, I, I, I, P, P],
    "craft_conv2d_nhwc2": [P, L, I, P, L, I, P, P, P, L, I, I, I, I, P, L, I, I, I, I, P],
    "craft_pack_operands": [P, I, P],
    "craft_wgrad_pk": [P, P, P, I, I, L, I, L, I, L, L, I, I, I, P, I, P],
    "craft_norm_act_fwd": [P, L, P, I, P, P, I, P, L, P, L, I, I, I, P],
    "craft_norm_act_bwd_reduce": [P, L, P, L, P, L, P, I, P, P, I, I, P, I, I, I, P],
    "craft_norm_act_bwd_apply": [P, L, P, L, P, L, P, I, P, P, I, I, P, I, P, L, P, L, I, I, I, P],
    "craft_bn_finalize": [P, I, I, ctypes.c_double, F, F, P, P, P, P],
    "craft_norm_bwd_finalize": [P, I, I, ctypes.c_double, I, P, P, P, P],
    "craft_stem_im2col": [P, I, I, I, P, P],
    "craft_zero_stuff2": [P, L, I, I, I, I, P, L, P],
    "craft_colsum": [P, L, L, I, P, P],
    "craft_act_fwd": [P, L, P, L, L, I, I, F, P],
    "craft_act_bwd": [P, L, P, L, P, L, L, I, I, F, P],
    "craft_act_bwd2": [P, L, P, L, P, L, P, L, L, I, I, F, I, P],
    "craft_dropout": [P, P, L, F, ctypes.c_ulonglong, P],
    "craft_tokens_bwd": [P, L, P, L, P, L, L, I, I, I, P],
    "craft_attn_softmax_fwd": [P, L, I, I, I, I, P, I, F, I, P, P, P, F, ctypes.c_ulonglong, P, L, I, I, P],
    "craft_attn_softmax_bwd": [P, P, L, I, I, I, I, I, F, P, P, P, F, ctypes.c_ulonglong, P, L, I, I, P],
    "craft_relpos_add": [P, L, I, I, I, P, L, P, L, F, P],
    "craft_relpos_bwd": [P, L, I, I, I, P, L, I, P, L, I, F, P],
    "craft_reduce_replicas": [P, I, I, P, P],
    "craft_corr_pool_fwd": [P, L, I, I, I, I, P, I, F, P, P, P, P, P],
    "craft_corr_lookup_bwd": [P, L, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "craft_corr_pyramid_bwd": [P, P, P, P, P, P, I, I, I, P, P],
    "craft_corr_pool_bwd": [P, L, I, I, I, I, P, I, F, P, P, P, P, P, P, I, P, P, P],
    "craft_mode_pool_ln_bwd": [P, P, L, P, P, P, L, I, I, I, I, P, P, L, P, P],
    "craft_convex_upsample_bwd": [P, L, P, P, I, I, I, P, L, P, L, P],
    "craft_flow_tokens": [P, P, L, P, P, P, P],
    "craft_gru_zr_fwd": [P, L, P, L, P, P, P, L, I, P],
    "craft_gru_out_fwd": [P, L, P, P, L, P, P, L, L, I, P],
    "craft_gru_out_bwd": [P, L, P, P, P, L, P, P, P, L, I, P, P],
    "craft_gru_zr_bwd": [P, P, L, P, P, P, L, P, P, L, I, P, P, L, P],
    "craft_coords_init": [P, I, I, I, P, P, P, P],
    # ---- input pipeline
    "craft_aug_spatial": [P, I, I, I, I, F, F, I, I, I, I, I, I, I, P, P],
    "craft_aug_photo": [P, L, I, F, F, P],
    "craft_aug_sparse": [P, P, I, I, F, F, I, I, I, I, I, P, P, P, P],
    "craft_aug_erase": [P, I, I, P, I, F, F, F, P],
    "craft_aug_shift": [P, P, P, I, I, I, I, P, P, P, P, P],
    "craft_aug_blur": [P, I, I, I, I, F, P, P],
}


# host-side helpers of the library (no stream argument, host pointers): bound where they are used
HOST_FUNCTIONS = {"craft_png_unfilter", "craft_pack_conv_job_bytes", "craft_pack_conv_job_fill"}

# Named policies.  "mixed" is the default for mixed_precision=True: split-fp16 (F16X3: fp32 operands as hi + lo fp16 planes,
# 3 fp16 MFMAs per product, fp32 accumulate -- fp32-class results) for the projections, Q K^T and every convolution, and
# plain fp16 storage + fp16 MFMA for the attention probabilities (P V); measured mean end-point deviation from the fp32
# path < 1e-3 px at 448x1024 / 12 iters.  "mixed_fp32conv": fp16 attention contractions + exact
# fp32 MFMA convolutions.
NAMED_POLICIES = {"mixed": "proj=f16x3,score=f16x3,pv=fp16,conv=f16x3,wgx=fp16,wgy=fp16,dxw=fp16,sbw=fp16", "mixed_fp32conv": "proj=fp16,score=fp16,pv=fp16,conv=fp32",
                  # training (activations and probabilities stay fp32 in memory; the roles select the MFMA operand mode of the
                  # forward AND backward contractions): everything fp32-class / bf16 MFMA for the cross- and self-attention
                  # contractions (BASELINE.json configs[4]: "bf16 MFMA cross-attention")
                  "train_f16x3": "proj=f16x3,score=f16x3,pv=f16x3,conv=f16x3",
                  "train_bf16attn": "proj=f16x3,score=bf16,pv=bf16,conv=f16x3,wgx=fp16,wgy=fp16,dxw=fp16",
                  # + the two CNN encoders with bf16 MFMA operands (the reference's --mixed_precision runs fnet / cnet under fp16
                  # autocast, network.py:179-183; bf16 so that no loss scaling is needed).  With args.hip_encoders=False the
                  # PyTorch-ROCm modules run under torch.autocast(bfloat16) instead
                  "train_bf16": "proj=f16x3,score=bf16,pv=bf16,conv=f16x3,enc=bf16",
                  # bf16 MFMA operands everywhere, fp32 accumulation / activations / master weights: the precision class of the
                  # reference's own training runs (every shipped train-*.sh passes --mixed_precision: fp16 autocast + GradScaler)
                  "train_amp_bf16": "proj=bf16,score=bf16,pv=bf16,conv=bf16,enc=bf16",
                  # the reference's own recipe: fp16 MFMA operands everywhere (fp16 autocast, train.py:215) -- legal only under a loss
                  # scale (train.Trainer's counterpart of GradScaler, train.py:231-238); without one forward_train promotes fp16 to f16x3
                  "train_amp_fp16": "proj=fp16,score=fp16,pv=fp16,conv=fp16,enc=fp16"}


class CraftHipError(RuntimeError):
    pass


# Re-laid-out weight caches (packed conv weights, BatchNorm-folded encoder weights, host copies of scalars) are keyed on
# (data_ptr, _version) of their source parameters AND on this counter: an optimizer that updates parameters through raw
# pointers (train.FlatAdamW's fused kernel) changes neither, so it bumps the epoch instead.
_weights_epoch = [0]


def weights_epoch() -> int:
    return _weights_epoch[0]


def bump_weights_epoch() -> None:
    _weights_epoch[0] += 1


class Precision:
    """MFMA operand precision per role of the hot path.

    proj  : nn.Linear projections (Q, K, V)            score : Q K^T (correlation build, attention logits)
    pv    : attention apply O = P V (also the storage type of the probabilities P)
    conv  : update-block convolutions (motion encoder, SepConvGRU, flow / mask heads)
    enc   : the two CNN encoders' convolutions (defaults to ``conv`` when not given)
    (training only) operand modes of the backward products of f16x3 layers; "fp16" = that operand rounded to ONE fp16 plane, default: the
    layer's mode (both planes, three MFMAs per product):
    wgx   : the activations X of the weight gradients dW = dY^T X          wgy : their gradient operand dY
            (wgx alone: 2 MFMAs, dW to ~2e-4 relative instead of ~2e-5; both: 1 MFMA, plain fp16 products under the loss scale.  Weight
            gradients are leaves of the backward graph: their rounding error does not propagate)
    dxw   : the weights W of the input gradients dX = dY W^T (2 MFMAs; dY of the dX chain always keeps both planes)
    sbw   : both operands of the score-gradient products dQ = dS K, dK = dS^T Q of an f16x3 attention ("fp16" / "bf16": dS leaves the
            softmax backward as a 16-bit pack and the products run on the packed engine; the scores themselves stay f16x3)
    Spec strings: "fp32" | "bf16" | "fp16" (all roles) or e.g. "score=bf16,pv=fp16,conv=fp32,proj=fp32"
    (unnamed roles default to fp32)."""
    __slots__ = ("proj", "score", "pv", "conv", "enc", "wgx", "wgy", "dxw", "sbw")
    BACKWARD_ROLES = ("wgx", "wgy", "dxw", "sbw")

    def __init__(self, proj=PREC_F32, score=PREC_F32, pv=PREC_F32, conv=PREC_F32, enc=None, wgx=None, wgy=None, dxw=None, sbw=None):
        self.proj, self.score, self.pv, self.conv = proj, score, pv, conv
        self.enc = conv if enc is None else enc
        self.wgx, self.wgy, self.dxw, self.sbw = wgx, wgy, dxw, sbw          # None: the operand in the mode of its layer

    @staticmethod
    def parse(spec) -> "Precision":
        if isinstance(spec, Precision):
            return spec
        if isinstance(spec, int):
            return Precision(spec, spec, spec, spec)
        spec = NAMED_POLICIES.get(spec.strip(), spec.strip())
        if "=" not in spec:
            v = PREC_NAMES[spec]
            return Precision(v, v, PREC_F16 if v == PREC_F16X3 else v, v)      # (inference stores P in the pv type: fp16 here)
        p = Precision()
        seen = set()
        for item in spec.split(","):
            k, v = item.split("=")
            if k.strip() not in Precision.__slots__:
                raise ValueError(f"unknown precision role {k!r}")
            setattr(p, k.strip(), PREC_NAMES[v.strip()])
            seen.add(k.strip())
        if "enc" not in seen:
            p.enc = p.conv
        for r in Precision.BACKWARD_ROLES:
            if r not in seen:
                setattr(p, r, None)
        return p

    def __repr__(self):
        inv = {0: "fp32", 1: "bf16", 2: "fp16", 3: "f16x3", None: "layer"}
        return ",".join(f"{k}={inv[getattr(self, k)]}" for k in self.__slots__)


def pick(prec, role: str) -> int:
    return prec if isinstance(prec, int) else getattr(prec, role)


_GROUP_OVERRIDE = None


def conv_group_prec(prec, group: str) -> int:
    """MFMA operand mode of one group of update-block convolutions ("gru", "menc", "fh", "mask"): the policy's ``conv`` role unless
    the developer override CRAFT_CONV_GROUPS="gru=fp16,mask=fp16,..." names the group (sensitivity study, tools/conv_group_sweep.sh)."""
    global _GROUP_OVERRIDE
    if _GROUP_OVERRIDE is None:
        _GROUP_OVERRIDE = {}
        for item in filter(None, os.environ.get("CRAFT_CONV_GROUPS", "").split(",")):
            k, v = item.split("=")
            _GROUP_OVERRIDE[k.strip()] = PREC_NAMES[v.strip()]
    cp = pick(prec, "conv")
    if cp == PREC_F32:
        return cp
    return _GROUP_OVERRIDE.get(group, cp)


def lib_path() -> str:
    return _LIB_PATH


def load():
    """dlopen the extension (once).  Raises if it has not been built (python -m craft_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CRAFT_HIP_LIB", _LIB_PATH)      # (override: A/B runs of two builds on one GPU box)
    if not os.path.exists(path):
        raise CraftHipError(f"{path} not found: build it with `python -m craft_amd.build` "
                            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(path)
    lib.craft_hip_abi_version.restype = c_int
    lib.craft_hip_error_string.restype = c_char_p
    lib.craft_hip_error_string.argtypes = [c_int]
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = sig
    if lib.craft_hip_abi_version() != ABI_VERSION:
        raise CraftHipError("libcraft_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise CraftHipError("craft_amd HIP ops need device tensors (no CPU fallback)")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Developer switch (tools/guard_run.py): CRAFT_HIP_DEBUG=<file> writes every entry point's name and arguments to <file> (rewound each call,
# so the file of a dead process holds the LAST call) and synchronises the device after the call, so an asynchronous GPU fault is raised
# inside the call that caused it.  Off (the default) the call path is untouched.
_DEBUG_PATH = os.environ.get("CRAFT_HIP_DEBUG", "")
_debug_file, _debug_n = None, [0]


def _debug_note(name, args, conv):
    global _debug_file
    if _debug_file is None:
        _debug_file = open(_DEBUG_PATH, "w")
    _debug_n[0] += 1
    desc = []
    for a, c in zip(args, conv):
        if isinstance(a, torch.Tensor):
            desc.append(f"T{tuple(a.shape)}:{str(a.dtype)[6:]}@{c:#x}+{a.numel() * a.element_size()}/{a.untyped_storage().nbytes() - (c - a.untyped_storage().data_ptr())}")
        else:
            desc.append(repr(c))
    _debug_file.seek(0)
    _debug_file.write(f"#{_debug_n[0]} {name}({', '.join(desc)})\n")
    _debug_file.truncate()
    _debug_file.flush()


def call(name: str, *args):
    lib = load()
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor) or a is None:
            conv.append(_ptr(a))
        else:
            conv.append(a)
    if _DEBUG_PATH:
        _debug_note(name, args, conv)
    rc = getattr(lib, name)(*conv, _stream())
    if rc != 0:
        msg = lib.craft_hip_error_string(rc).decode()
        raise CraftHipError(f"{name} failed with code {rc}: {msg}")
    if _DEBUG_PATH:
        torch.cuda.synchronize()


_CARRAY_TYPES = {}


def carray(ctype, values):
    """A ctypes array of ``values`` -- the array TYPE cached per (element type, length): ``ctype * n`` builds a new class object with a
    reference cycle on every evaluation, hundreds per training step that only the cyclic collector frees."""
    n = len(values)
    t = _CARRAY_TYPES.get((ctype, n))
    if t is None:
        t = _CARRAY_TYPES[(ctype, n)] = ctype * n
    return t(*values)


class ZeroPool:
    """The many small zero-initialised buffers of a training step (gradient accumulators, statistics replicas: ~300 torch.zeros, a 4 us
    fill kernel each) from ONE zeroed allocation per step.  The first step records the (shape, dtype) sequence; later steps allocate the
    whole sequence at ``begin`` (one memset) and hand the pieces out in order -- any deviation from the recorded sequence falls back to
    torch.zeros for the rest of the step and re-records.  Every piece is handed out once and the flat buffer is fresh per step, so
    nothing aliases across steps; a piece that outlives its step only keeps that step's flat buffer alive."""

    def __init__(self):
        self.plan, self.rec, self.flat, self.idx, self.ok = None, [], None, 0, False

    def begin(self, device):
        if self.rec and (self.plan is None or not self.ok):
            self.plan = self.rec                        # (re-)learned from the step that just ran
        self.rec, self.idx, self.ok, self.flat = [], 0, False, None
        if self.plan:
            total = sum(n for _, _, _, n in self.plan)
            self.flat = torch.zeros(total, device=device, dtype=torch.uint8)
            self.ok = True
            self.off = 0

    def zeros(self, shape, device, dtype=torch.float32):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        numel = 1
        for s_ in shape:
            numel *= s_
        live = numel * _ELEM_SIZE[dtype]
        nbytes = (live + 255) // 256 * 256
        self.rec.append((shape, dtype, None, nbytes))
        if self.ok:
            if self.idx < len(self.plan) and self.plan[self.idx][:2] == (shape, dtype) and self.flat.device.type == torch.device(device).type:
                out = self.flat[self.off:self.off + live].view(dtype).view(shape)
                self.off += nbytes
                self.idx += 1
                return out
            self.ok = False                             # the sequence changed (other shapes, another model): plain allocations from here on
        return torch.zeros(shape, device=device, dtype=dtype)


_ELEM_SIZE = {torch.float32: 4, torch.float64: 8, torch.int32: 4, torch.int64: 8, torch.int16: 2, torch.uint8: 1, torch.float16: 2, torch.bfloat16: 2}
_POOL = [None]
_NO_ZERO_POOL = bool(os.environ.get("CRAFT_NO_ZERO_POOL"))      # developer switch: every piece its own allocation (guard-page runs)


def set_zero_pool(pool):
    """Make ``pool`` (or None) the source of ``zeros`` until the next call (train_forward.forward_train: one pool per model)."""
    _POOL[0] = pool


def zeros(shape, device, dtype=torch.float32):
    """A zero-initialised per-step temporary (see ZeroPool); plain torch.zeros outside a training step."""
    pool = None if _NO_ZERO_POOL else _POOL[0]
    return pool.zeros(shape, device, dtype) if pool is not None else torch.zeros(shape, device=device, dtype=dtype)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m
