"""ctypes binding of libunetmi.so (C ABI declared in include/unetmi.h).

The library is the only compute path of this package: importing this module
fails loudly when the shared object is missing or does not export a symbol
the header declares.  There is no CPU or eager fallback.
"""

import ctypes
import os

import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunetmi.so")
# Developer probes (tools/abl/run.sh) time instrumented builds of the library; they point at one here instead of
# overwriting the product file.  Never silent: the override is announced on stderr.
if os.environ.get("UNETMI_LIBRARY"):
    LIB_PATH = os.path.abspath(os.environ["UNETMI_LIBRARY"])
    sys.stderr.write("unet_amd: loading {} (UNETMI_LIBRARY override)\n".format(LIB_PATH))

# activation / input-mode codes of include/unetmi.h
ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SILU, ACT_SOFTPLUS, ACT_SIGMOID, ACT_TANH, ACT_GELU = range(8)
IN_SINGLE, IN_CONCAT, IN_ADD = range(3)
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
B16_OUT_NCHW_F32, B16_IN_F16, B16_OUT_F16 = 1, 2, 4      # bits of UnetmiConvDesc.reserved[2] for the 16-bit convolution

EXPORTED_SYMBOLS = (
    "unetmi_version",
    "unetmi_last_error",
    "unetmi_conv2d_packed_weight_bytes",
    "unetmi_conv2d_kernel_name",
    "unetmi_conv2d_pack_weights",
    "unetmi_conv2d_fwd",
    "unetmi_conv2d_fwd_stats_workspace_bytes",
    "unetmi_conv2d_fwd_stats",
    "unetmi_dwconv2d_fwd",
    "unetmi_dwconv2d_wgrad_workspace_bytes",
    "unetmi_dwconv2d_wgrad",
    "unetmi_layernorm_f_fwd",
    "unetmi_tanh_mask_mul",
    "unetmi_crf_apply",
    "unetmi_bn_workspace_bytes",
    "unetmi_bn_train_stats",
    "unetmi_bn_train_stats_static",
    "unetmi_bn_act_fwd",
    "unetmi_bn_act_bwd",
    "unetmi_conv2d_wgrad_workspace_bytes",
    "unetmi_conv2d_wgrad",
    "unetmi_adam_step",
    "unetmi_resblock_layer_fwd",
    "unetmi_pwconv64_fwd",
    "unetmi_lstm_recurrence",
    "unetmi_lstm_recurrence_xcd_workspace_bytes",
    "unetmi_lstm_recurrence_xcd",
    "unetmi_lstm_recurrence_xcd_train",
    "unetmi_lstm_bptt_xcd_workspace_bytes",
    "unetmi_lstm_bptt_xcd",
    "unetmi_gru_recurrence",
    "unetmi_gru_recurrence_train",
    "unetmi_gru_gate_fwd",
    "unetmi_gru_bptt",
    "unetmi_glu_fwd",
    "unetmi_instnorm_act_fwd",
    "unetmi_layernorm_w_act_fwd",
    "unetmi_conv2d_b16_packed_weight_bytes",
    "unetmi_conv2d_b16_kernel_name",
    "unetmi_conv2d_b16_pack_weights",
    "unetmi_conv2d_b16_fwd",
    "unetmi_conv2d_b16_head_fwd",
    "unetmi_conv2d_b16_nchw_fwd",
    "unetmi_blocked_from_nchw_sum",
    "unetmi_blocked_from_nchw",
    "unetmi_nchw_from_blocked",
    "unetmi_bn_running_update",
    "unetmi_gather_f32",
    "unetmi_comm_unique_id",
    "unetmi_comm_init",
    "unetmi_comm_world",
    "unetmi_allreduce_f32",
    "unetmi_comm_destroy",
)


class ConvDesc(ctypes.Structure):
    """struct UnetmiConvDesc."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "C0", "C1", "H", "W", "K", "OH", "OW", "kh", "kw", "sh", "sw", "ph", "pw",
        "dh", "dw", "oph", "opw", "transposed", "in_mode", "act", "act_first",
        "act_param_per_channel", "has_gate", "has_residual")] + [("reserved", ctypes.c_int32 * 3)]


E_BAD_DESC, E_BAD_ARG, E_HIP, E_UNSUPPORTED = -1, -2, -3, -4


class UnetmiError(RuntimeError):
    def __init__(self, message, code=0):
        super().__init__(message)
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "unet_amd: {} is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C unet_amd/csrc`; there is no fallback path".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for sym in EXPORTED_SYMBOLS:
        if not hasattr(lib, sym):
            raise ImportError("unet_amd: {} does not export {}".format(LIB_PATH, sym))
    vp, fp = ctypes.c_void_p, ctypes.c_void_p
    lib.unetmi_version.restype = ctypes.c_int
    lib.unetmi_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.unetmi_last_error.restype = None
    lib.unetmi_conv2d_packed_weight_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.unetmi_conv2d_packed_weight_bytes.restype = ctypes.c_size_t
    lib.unetmi_conv2d_kernel_name.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_char_p, ctypes.c_size_t]
    lib.unetmi_conv2d_kernel_name.restype = ctypes.c_int
    lib.unetmi_conv2d_pack_weights.argtypes = [ctypes.POINTER(ConvDesc), fp, fp]
    lib.unetmi_conv2d_pack_weights.restype = ctypes.c_int
    lib.unetmi_conv2d_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [fp] * 10 + [vp]
    lib.unetmi_conv2d_fwd.restype = ctypes.c_int
    lib.unetmi_conv2d_fwd_stats_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.unetmi_conv2d_fwd_stats_workspace_bytes.restype = ctypes.c_size_t
    lib.unetmi_conv2d_fwd_stats.argtypes = ([ctypes.POINTER(ConvDesc)] + [fp] * 7 + [ctypes.c_float, fp, fp, vp, ctypes.c_size_t, vp])
    lib.unetmi_conv2d_fwd_stats.restype = ctypes.c_int
    ci, cf = ctypes.c_int, ctypes.c_float
    lib.unetmi_dwconv2d_fwd.argtypes = [ci] * 10 + [fp] * 5 + [ci, vp]
    lib.unetmi_dwconv2d_wgrad_workspace_bytes.argtypes = [ci] * 3
    lib.unetmi_dwconv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.unetmi_dwconv2d_wgrad.argtypes = [ci] * 10 + [fp] * 5 + [vp, ctypes.c_size_t, vp]
    lib.unetmi_dwconv2d_wgrad.restype = ctypes.c_int
    lib.unetmi_layernorm_f_fwd.argtypes = [ci] * 6 + [cf] + [fp] * 4 + [ci, vp]
    lib.unetmi_tanh_mask_mul.argtypes = [ci] * 5 + [cf] + [fp] * 3 + [ci, vp]
    lib.unetmi_crf_apply.argtypes = [ci] * 7 + [fp] * 3 + [vp]
    sz, ll = ctypes.c_size_t, ctypes.c_longlong
    lib.unetmi_bn_workspace_bytes.argtypes = [ci] * 3
    lib.unetmi_bn_workspace_bytes.restype = sz
    lib.unetmi_bn_train_stats.argtypes = [ci] * 3 + [fp, cf, cf] + [fp] * 4 + [vp, sz, vp]
    lib.unetmi_bn_train_stats_static.argtypes = [ci] * 3 + [fp, cf, cf] + [fp] * 4 + [vp]
    lib.unetmi_bn_train_stats_static.restype = ctypes.c_int
    lib.unetmi_bn_act_fwd.argtypes = [ci] * 3 + [fp] * 5 + [ci, fp, fp, vp]
    lib.unetmi_bn_act_bwd.argtypes = [ci] * 3 + [fp, ll] + [fp] * 5 + [ci, fp, ci] + [fp] * 4 + [vp, sz, vp]
    lib.unetmi_conv2d_wgrad_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.unetmi_conv2d_wgrad_workspace_bytes.restype = sz
    lib.unetmi_conv2d_wgrad.argtypes = [ctypes.POINTER(ConvDesc)] + [fp] * 4 + [vp, sz, vp]
    lib.unetmi_resblock_layer_fwd.argtypes = [ci] * 5 + [fp] * 9 + [ci, fp, ci, vp]
    lib.unetmi_resblock_layer_fwd.restype = ctypes.c_int
    lib.unetmi_pwconv64_fwd.argtypes = [ci] * 3 + [fp] * 7 + [ci, ci, fp, ci, vp]
    lib.unetmi_pwconv64_fwd.restype = ctypes.c_int
    lib.unetmi_lstm_recurrence.argtypes = [ci] * 3 + [fp] * 3 + [vp, vp]
    lib.unetmi_lstm_recurrence.restype = ctypes.c_int
    lib.unetmi_lstm_recurrence_xcd_workspace_bytes.argtypes = []
    lib.unetmi_lstm_recurrence_xcd_workspace_bytes.restype = ctypes.c_size_t
    lib.unetmi_lstm_recurrence_xcd.argtypes = [ci] * 3 + [fp] * 3 + [ci, vp, ctypes.c_size_t, vp]
    lib.unetmi_lstm_recurrence_xcd.restype = ctypes.c_int
    lib.unetmi_lstm_recurrence_xcd_train.argtypes = [ci] * 3 + [fp] * 4 + [vp, ctypes.c_size_t, vp]
    lib.unetmi_lstm_recurrence_xcd_train.restype = ctypes.c_int
    lib.unetmi_lstm_bptt_xcd_workspace_bytes.argtypes = []
    lib.unetmi_lstm_bptt_xcd_workspace_bytes.restype = ctypes.c_size_t
    lib.unetmi_lstm_bptt_xcd.argtypes = [ci] * 3 + [fp] * 4 + [vp, ctypes.c_size_t, vp]
    lib.unetmi_lstm_bptt_xcd.restype = ctypes.c_int
    lib.unetmi_gru_recurrence.argtypes = [ci] * 3 + [fp] * 3 + [ci, fp, vp, vp]
    lib.unetmi_gru_recurrence.restype = ctypes.c_int
    lib.unetmi_gru_recurrence_train.argtypes = [ci] * 3 + [fp] * 5 + [vp, vp]
    lib.unetmi_gru_recurrence_train.restype = ctypes.c_int
    lib.unetmi_gru_bptt.argtypes = [ci] * 3 + [fp] * 5 + [vp, vp]
    lib.unetmi_gru_bptt.restype = ctypes.c_int
    ll = ctypes.c_longlong
    lib.unetmi_gru_gate_fwd.argtypes = [ll, ci, fp, ll, fp, fp, fp, ll, fp, ll, vp]
    lib.unetmi_gru_gate_fwd.restype = ctypes.c_int
    lib.unetmi_glu_fwd.argtypes = [ci] * 3 + [fp] * 3 + [ci] + [fp] * 3 + [ci, vp]
    lib.unetmi_glu_fwd.restype = ctypes.c_int
    lib.unetmi_instnorm_act_fwd.argtypes = [ci] * 3 + [fp] * 3 + [cf, ci, fp, ci, fp, fp, ci, vp]
    lib.unetmi_instnorm_act_fwd.restype = ctypes.c_int
    lib.unetmi_layernorm_w_act_fwd.argtypes = [ci] * 4 + [fp] * 3 + [cf, ci, fp, ci, fp, fp, ci, vp]
    lib.unetmi_layernorm_w_act_fwd.restype = ctypes.c_int
    lib.unetmi_adam_step.argtypes = [sz] + [fp] * 4 + [cf] * 6 + [ci, vp]
    lib.unetmi_conv2d_b16_packed_weight_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.unetmi_conv2d_b16_packed_weight_bytes.restype = sz
    lib.unetmi_conv2d_b16_kernel_name.argtypes = [ctypes.POINTER(ConvDesc), ci, ctypes.c_char_p, sz]
    lib.unetmi_conv2d_b16_kernel_name.restype = ctypes.c_int
    lib.unetmi_conv2d_b16_pack_weights.argtypes = [ctypes.POINTER(ConvDesc), fp, fp]
    lib.unetmi_conv2d_b16_pack_weights.restype = ctypes.c_int
    lib.unetmi_conv2d_b16_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [fp] * 9 + [vp]
    lib.unetmi_conv2d_b16_fwd.restype = ctypes.c_int
    lib.unetmi_conv2d_b16_nchw_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [fp] * 10 + [vp]
    lib.unetmi_conv2d_b16_nchw_fwd.restype = ctypes.c_int
    lib.unetmi_blocked_from_nchw_sum.argtypes = [ci] * 3 + [fp, fp, ci, fp, ci, vp]
    lib.unetmi_blocked_from_nchw_sum.restype = ctypes.c_int
    lib.unetmi_conv2d_b16_head_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [fp] * 10 + [ci] * 3 + [fp, vp]
    lib.unetmi_conv2d_b16_head_fwd.restype = ctypes.c_int
    lib.unetmi_blocked_from_nchw.argtypes = [ci] * 3 + [fp, ci, fp, vp]
    lib.unetmi_blocked_from_nchw.restype = ctypes.c_int
    lib.unetmi_nchw_from_blocked.argtypes = [ci] * 3 + [fp, ci, fp, ci, vp]
    lib.unetmi_nchw_from_blocked.restype = ctypes.c_int
    lib.unetmi_bn_running_update.argtypes = [ci, ll, cf, cf, fp, fp, fp, fp, vp]
    lib.unetmi_bn_running_update.restype = ctypes.c_int
    lib.unetmi_gather_f32.argtypes = [sz, fp, fp, fp, vp]
    lib.unetmi_gather_f32.restype = ctypes.c_int
    lib.unetmi_comm_unique_id.argtypes = [vp]
    lib.unetmi_comm_init.argtypes = [ci, ci, vp]
    lib.unetmi_comm_world.argtypes = []
    lib.unetmi_allreduce_f32.argtypes = [fp, sz, vp]
    lib.unetmi_comm_destroy.argtypes = []
    for f in (lib.unetmi_comm_unique_id, lib.unetmi_comm_init, lib.unetmi_comm_world, lib.unetmi_allreduce_f32,
              lib.unetmi_comm_destroy):
        f.restype = ctypes.c_int
    for f in (lib.unetmi_bn_train_stats, lib.unetmi_bn_act_fwd, lib.unetmi_bn_act_bwd, lib.unetmi_conv2d_wgrad,
              lib.unetmi_adam_step):
        f.restype = ctypes.c_int
    for f in (lib.unetmi_dwconv2d_fwd, lib.unetmi_layernorm_f_fwd, lib.unetmi_tanh_mask_mul, lib.unetmi_crf_apply):
        f.restype = ctypes.c_int
    return lib


lib = _load()


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib.unetmi_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(code, what):
    if code != 0:
        raise UnetmiError("{} failed ({}): {}".format(what, code, last_error()), code)
