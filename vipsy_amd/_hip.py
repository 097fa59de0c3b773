"""ctypes binding of include/vipsy_amd.h (the C-ABI shared library built from vipsy_amd/csrc).

The product path has NO fallback: if libvipsy_hip.so is missing or an entry point fails, this raises.
torch is used only as the owner of device memory and streams (plumbing).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VX_LIB names another build of the same sources (tests/test_gpu_schedules.py runs the oracle comparisons against a library
# compiled with a different instruction schedule: a hazard that only one schedule hides must show in the other)
LIB_PATH = os.environ.get("VX_LIB") or os.path.join(_HERE, "_lib", "libvipsy_hip.so")
ABI_VERSION = 5         # include/vipsy_amd.h: VX_ABI_VERSION (struct layouts and argument lists this binding was written for)


class VxError(RuntimeError):
    pass


class IrtCfg(ctypes.Structure):
    """struct vx_irt_cfg (include/vipsy_amd.h)."""
    _fields_ = [("model", ctypes.c_int32), ("D", ctypes.c_int32), ("J", ctypes.c_int32), ("H", ctypes.c_int32),
                ("Dc", ctypes.c_float), ("scale", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_uint32), ("stream", ctypes.c_uint32), ("step_dev", ctypes.c_void_p),
                ("rows_ring", ctypes.c_void_p), ("rows_ring_stride", ctypes.c_int64), ("rows_ring_slots", ctypes.c_int32),
                ("_pad", ctypes.c_int32)]


class AdamSeg(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int64), ("end", ctypes.c_int64), ("lr", ctypes.c_float), ("_pad", ctypes.c_float)]


class AdamTail(ctypes.Structure):
    """struct vx_adam_tail (include/vipsy_amd.h)."""
    _fields_ = [("pA", ctypes.c_void_p), ("mA", ctypes.c_void_p), ("vA", ctypes.c_void_p), ("freeA", ctypes.c_void_p),
                ("nA", ctypes.c_int64), ("segsA", ctypes.POINTER(AdamSeg)),
                ("pB", ctypes.c_void_p), ("gB", ctypes.c_void_p), ("mB", ctypes.c_void_p), ("vB", ctypes.c_void_p),
                ("nB", ctypes.c_int64), ("segsB", ctypes.POINTER(AdamSeg)),
                ("n_segsA", ctypes.c_int32), ("n_segsB", ctypes.c_int32), ("t", ctypes.c_int32),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("loss_ring", ctypes.c_void_p)]


class HoDinaCfg(ctypes.Structure):
    """struct vx_hodina_cfg (include/vipsy_amd.h)."""
    _fields_ = [("K", ctypes.c_int32), ("J", ctypes.c_int32), ("H", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("scale", ctypes.c_float), ("_pad2", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_uint32), ("stream", ctypes.c_uint32), ("step_dev", ctypes.c_void_p)]


_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int32
_U32 = ctypes.c_uint32
_U64 = ctypes.c_uint64
_F = ctypes.c_float
_CFG = ctypes.POINTER(IrtCfg)

# name -> (restype, argtypes); every symbol the header declares must be listed here
SIGNATURES = {
    "vx_abi_version": (ctypes.c_int, []),
    "vx_build_info": (ctypes.c_char_p, []),
    "vx_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "vx_prof_count": (ctypes.c_int, []),
    "vx_prof_read": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ctypes.c_int)]),
    "vx_prof_units": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "vx_philox_normals": (ctypes.c_int, [_P, _P, _I64, _I64, _I32, _U64, _U32, _U32, _P]),
    "vx_philox_raw": (ctypes.c_int, [_P, _I64, _I64, _U64, _U32, _U32, _P]),
    "vx_mvn_enc_forward": (ctypes.c_int, [_CFG, _P, _P, _I64, _I64] + [_P] * 6 + [_P] + [_P] * 5 + [_P, _P] + [_P, _P, _P, _P]),
    "vx_mvn_pack_floats": (_I64, [_CFG]),
    "vx_mvn_pack_opmax_offset": (_I64, [_CFG, _I64]),
    "vx_irt_lik_workspace_floats": (_I64, [_CFG, _I64]),
    "vx_irt_lik_ximg_bytes": (_I64, [_CFG, _I64]),
    "vx_irt_lik_grad": (ctypes.c_int, [_CFG, _P, _P, _I64] + [_P] * 5 + [_P] * 4 + [_P, _P, _I64, _P] + [_P] * 3 + [_P, _P]),
    "vx_mvn_enc_param_floats": (_I64, [_CFG]),
    "vx_mvn_enc_bwd_workspace_floats": (_I64, [_CFG, _I64]),
    "vx_mvn_enc_bwd_layout": (ctypes.c_int, [_CFG, _I64]),
    "vx_mvn_enc_backward": (ctypes.c_int, [_CFG, _P, _P, _I64] + [_P] * 6 + [_P] * 3 + [_P, _I64] + [_P, _P, _P, _I32, _P]),
    "vx_mvn_enc_backward_loss": (ctypes.c_int, [_CFG, _P, _P, _I64] + [_P] * 6 + [_P] * 3 + [_P, _I64] + [_P, _P, _P, _I32] +
                                 [_P, _P, _F, _P, _P, _P]),
    "vx_mvn_enc_bwd_gd_offset": (_I64, [_CFG, _I64]),
    "vx_mvn_enc_bwd_hs_offset": (_I64, [_CFG, _I64]),
    "vx_irt1d_workspace_floats": (_I64, [_CFG, _I64]),
    "vx_irt1d_grad": (ctypes.c_int, [_CFG, _P, _P, _I64, _I64] + [_P] * 3 + [_P] * 4 + [_P] * 4 + [_P, _P, _P, _P]),
    "vx_irt1d_sparse_workspace_floats": (_I64, [_CFG, _I64]),
    "vx_irt1d_sparse_grad": (ctypes.c_int, [_CFG, _P, _P, _I32, _P, _I64, _I64] + [_P] * 3 + [_P] * 4 + [_P] * 4 + [_P, _P, _P, _P]),
    "vx_irt1d_grad_adam": (ctypes.c_int, [_CFG, _P, _P, _I64, _I64] + [_P] * 3 + [_P] * 4 + [_P] * 4 + [_P, _P, _P, ctypes.POINTER(AdamTail), _P]),
    "vx_irt1d_sparse_grad_adam": (ctypes.c_int, [_CFG, _P, _P, _I32, _P, _I64, _I64] + [_P] * 3 + [_P] * 4 + [_P] * 4 +
                                  [_P, _P, _P, ctypes.POINTER(AdamTail), _P]),
    "vx_mvn_bbvi_forward": (ctypes.c_int, [_CFG, _I64, _P, _I64, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "vx_mvn_bbvi_bwd_workspace_floats": (_I64, [_CFG, _I64, _I32]),
    "vx_mvn_bbvi_backward": (ctypes.c_int, [_CFG, _I64, _P, _P, _I32, _P, _P, _P, _P, _P, _P]),
    "vx_norm_enc_param_floats": (_I64, [_CFG]),
    "vx_norm_enc_pack_floats": (_I64, [_CFG]),
    "vx_norm_enc_forward": (ctypes.c_int, [_CFG, _P, _P, _I64] + [_P] * 6 + [_P] * 3 + [_P, _P]),
    "vx_norm_enc_bwd_workspace_floats": (_I64, [_CFG, _I64]),
    "vx_norm_enc_backward": (ctypes.c_int, [_CFG, _P, _P, _I64] + [_P] * 5 + [_P, _I64] + [_P, _P, _P]),
    "vx_hodina_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_hodina_grad": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _P, _P, _I64, _I64] + [_P] * 3 + [_P] * 5 + [_P] * 4 + [_P, _P]),
    "vx_ccdm_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_ccdm_grad": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _I32, _P, _P, _I64] + [_P] * 3 + [_P, _P, _P, _P]),
    "vx_cdm_sf_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_cdm_sf_grad": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _I32, _I32, _F, _P, _P, _I64, _I64] + [_P] * 3 + [_P, _P, _P, _F, _I32] + [_P] * 3 + [_P, _P, _P]),
    "vx_loo_baseline": (ctypes.c_int, [_P, _I32, _I64, _I32, _P, _P]),
    "vx_irt1d_score_grad": (ctypes.c_int, [_I64, ctypes.c_float, _P, _P, _P, _P, _P, ctypes.c_float, _I32, _P, _P, _P, _P]),
    "vx_mvn_score_operands": (ctypes.c_int, [_CFG, _I64, _P, _I32] + [_P] * 8 + [_F, _I32] + [_P] * 6),
    "vx_mvn_score_diag": (ctypes.c_int, [_CFG, _I64, _P, _P, _I32, _P, _P]),
    "vx_mvn_score_heads_workspace_floats": (_I64, [_CFG]),
    "vx_mvn_score_heads": (ctypes.c_int, [_CFG, _I64, _P] + [_P] * 7 + [_P, _F, _I32] + [_P] * 5),
    "vx_bin_enc_param_floats": (_I64, [ctypes.POINTER(HoDinaCfg)]),
    "vx_bin_enc_forward": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _P, _P, _I64] + [_P] * 4 + [_P, _P, _P]),
    "vx_bin_enc_bwd_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_bin_enc_backward": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _P, _P, _I64] + [_P] * 3 + [_P, _P, _P]),
    "vx_synth_irt": (ctypes.c_int, [_CFG, _I64, _I64] + [_P] * 5 + [_F, _P, _P, _P]),
    "vx_synth_cdm": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _I32, _I32, _F, _I64, _I64] + [_P] * 5 + [_F, _P, _P, _P, _P]),
    "vx_sm_enc_param_floats": (_I64, [ctypes.POINTER(HoDinaCfg)]),
    "vx_sm_enc_forward": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _P, _P, _I64] + [_P] * 4 + [_P, _P, _P]),
    "vx_col_reduce_workspace_floats": (_I64, [_I64, _I32]),
    "vx_col_reduce": (ctypes.c_int, [_I32, _P, _I64, _I32, _P, _P, _P, _P]),
    "vx_vaeccdm_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_vaeccdm_grad": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _I32, _P, _P, _I64] + [_P] * 3 + [_P, _P] + [_P, _P, _P, _P, _P]),
    "vx_sm_enc_bwd_workspace_floats": (_I64, [ctypes.POINTER(HoDinaCfg), _I64]),
    "vx_sm_enc_backward": (ctypes.c_int, [ctypes.POINTER(HoDinaCfg), _P, _P, _I64] + [_P] * 5 + [_P, _P, _P, _P]),
    "vx_reduce_slabs": (ctypes.c_int, [_P, _I64, _I64, _F, _P, _P]),
    "vx_sum_workspace_floats": (_I64, []),
    "vx_sum": (ctypes.c_int, [_P, _I64, _F, _P, _P, _P, _P]),
    "vx_sum2": (ctypes.c_int, [_P, _P, _I64, _F, _P, _P, _P, _P]),
    "vx_adam_step": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, ctypes.POINTER(AdamSeg), _I32, _I32, _P, _F, _F, _F, _P, _P, _P]),
    "vx_adam_step2": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, ctypes.POINTER(AdamSeg), _I32, _P, _P, _P, _P, _I64,
                                     ctypes.POINTER(AdamSeg), _I32, _I32, _P, _F, _F, _F, _P, _P, _P]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises VxError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VxError("HIP extension missing: %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback in the product path)" % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise VxError("cannot load %s: %s" % (LIB_PATH, e))
        try:
            handle.vx_abi_version.restype = ctypes.c_int
            abi = handle.vx_abi_version()
        except AttributeError:
            raise VxError("%s does not export vx_abi_version -- not a vipsy_amd library" % LIB_PATH)
        if abi != ABI_VERSION:
            raise VxError("%s has ABI %d, this binding expects %d -- rebuild (`make -C vipsy_amd/csrc`)" %
                          (LIB_PATH, abi, ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise VxError("%s lacks %s -- stale build, rebuild (`make -C vipsy_amd/csrc`)" % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise VxError("%s failed with code %d" % (what, rc))


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous(), "device buffers handed to the C ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise VxError("vipsy_amd needs an MI355X (HIP device); none is visible and there is no CPU fallback")
