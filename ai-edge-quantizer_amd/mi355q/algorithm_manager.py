"""Module-level algorithm registry with the GPU-backed algorithms registered.

Same surface as ref: algorithm_manager.py:43-75 (singleton + re-exported bound
methods + AlgorithmName) and the same registration pattern
(ref :150-164, 310-320, 365-383, 406-419, 437-451):
`functools.partial(materialize_fn, <alg>.get_tensor_quant_params)`.
Registered here, as there: the 50-op tables of min/max and OCTAV (the weight-bearing ops, the
virtual INPUT / OUTPUT ops and every activation-only op that static recipes quantize), MSE,
GPTQ, both Hadamard forms, OSCAR, float_casting and dequantized_weight_recovery -- all ten
algorithm keys, GPU backed.
"""
from __future__ import annotations

import enum
import functools

from . import algorithm_manager_api
from . import default_policy
from . import qtyping
from .algorithms.nonlinear_quantize import float_casting
from .algorithms.uniform_quantize import common_quantize
from .algorithms.uniform_quantize import dequantized_weight_recovery
from .algorithms.uniform_quantize import gptq
from .algorithms.uniform_quantize import hadamard_rotation
from .algorithms.uniform_quantize import mse
from .algorithms.uniform_quantize import naive_min_max_quantize
from .algorithms.uniform_quantize import octav
from .algorithms.uniform_quantize import oscar
from .utils import qsv_utils

_Op = qtyping.TFLOperationName

_alg_manager_instance = algorithm_manager_api.AlgorithmManagerApi()

get_quantization_func = _alg_manager_instance.get_quantization_func
get_supported_ops = _alg_manager_instance.get_supported_ops
get_update_qsv_func = _alg_manager_instance.get_update_qsv_func
get_init_qsv_func = _alg_manager_instance.get_init_qsv_func
register_op_quant_config_validation_func = (
    _alg_manager_instance.register_op_quant_config_validation_func)
register_config_check_policy_func = _alg_manager_instance.register_config_check_policy
register_quantized_op = _alg_manager_instance.register_quantized_op
is_op_registered = _alg_manager_instance.is_op_registered
is_algorithm_registered = _alg_manager_instance.is_algorithm_registered
check_op_quantization_config = _alg_manager_instance.check_op_quantization_config


class AlgorithmName(str, enum.Enum):
  """ref :65-75 (keys of algorithms outside the hot path are kept for recipe compatibility)."""
  NO_QUANTIZE = "no_quantize"
  MIN_MAX_UNIFORM_QUANT = naive_min_max_quantize.ALGORITHM_KEY
  FLOAT_CASTING = float_casting.ALGORITHM_KEY
  DEQUANTIZED_WEIGHT_RECOVERY = dequantized_weight_recovery.ALGORITHM_KEY
  OCTAV = octav.ALGORITHM_KEY
  HADAMARD_ROTATION = hadamard_rotation.CUSTOM_OP_ALGORITHM_KEY
  DECOMPOSED_HADAMARD_ROTATION = hadamard_rotation.DECOMPOSED_ALGORITHM_KEY
  MSE = mse.ALGORITHM_KEY
  GPTQ = gptq.ALGORITHM_KEY
  OSCAR = oscar.ALGORITHM_KEY


_MATERIALIZERS = {
    _Op.INPUT: common_quantize.materialize_input,
    _Op.OUTPUT: common_quantize.materialize_output,
    _Op.FULLY_CONNECTED: common_quantize.materialize_fc_conv,
    _Op.CONV_2D: common_quantize.materialize_fc_conv,
    _Op.DEPTHWISE_CONV_2D: common_quantize.materialize_fc_conv,
    _Op.EMBEDDING_LOOKUP: common_quantize.materialize_embedding_lookup,
    _Op.BATCH_MATMUL: common_quantize.materialize_batch_matmul,
    _Op.CONV_2D_TRANSPOSE: common_quantize.materialize_conv2d_transpose,
    # ops that only carry activations: quantized under static (SRQ) recipes
    _Op.SOFTMAX: common_quantize.materialize_softmax_and_logistic,
    _Op.LOGISTIC: common_quantize.materialize_softmax_and_logistic,
    _Op.TANH: common_quantize.materialize_tanh,
    _Op.STABLEHLO_COMPOSITE: common_quantize.materialize_composite,
}
for _name in ("RESHAPE", "AVERAGE_POOL_2D", "TRANSPOSE", "GELU", "ADD", "SUB", "MUL", "MEAN", "RSQRT",
              "CONCATENATION", "STRIDED_SLICE", "SPLIT", "SLICE", "SUM", "SELECT", "SELECT_V2",
              "DYNAMIC_UPDATE_SLICE", "PAD", "SQUARED_DIFFERENCE", "MAX_POOL_2D", "RESIZE_BILINEAR",
              "RESIZE_NEAREST_NEIGHBOR", "GATHER_ND", "PACK", "UNPACK", "DIV", "BROADCAST_TO", "SQRT",
              "GATHER", "HARD_SWISH", "MAXIMUM", "PADV2", "REDUCE_MIN", "EQUAL", "NOT_EQUAL",
              "MIRROR_PAD", "SPACE_TO_DEPTH", "RELU"):
  _MATERIALIZERS[_Op[_name]] = getattr(common_quantize, "materialize_" + _name.lower())


def _register_weight_algorithm(name, module, ops, calibration_func, update_qsv_func):
  register_op_quant_config_validation_func(name, common_quantize.check_op_quantization_config)
  register_config_check_policy_func(name, default_policy.DEFAULT_CONFIG_CHECK_POLICY)
  for op in ops:
    register_quantized_op(
        name, op, naive_min_max_quantize.init_qsvs, calibration_func=calibration_func,
        materialize_func=functools.partial(_MATERIALIZERS[op], module.get_tensor_quant_params),
        update_qsv_func=update_qsv_func)


# min/max: every op above (ref :78-164)
_register_weight_algorithm(AlgorithmName.MIN_MAX_UNIFORM_QUANT, naive_min_max_quantize,
                           list(_MATERIALIZERS), naive_min_max_quantize.min_max_calibrate,
                           qsv_utils.moving_average_update)
# OCTAV (ref :237-320): the min/max op table, min/max calibration
_register_weight_algorithm(AlgorithmName.OCTAV, octav, list(_MATERIALIZERS),
                           naive_min_max_quantize.min_max_calibrate, qsv_utils.moving_average_update)
# MSE (ref :385-419): weights of matmul / convolution style ops only
_register_weight_algorithm(AlgorithmName.MSE, mse,
                           [_Op.FULLY_CONNECTED, _Op.EMBEDDING_LOOKUP, _Op.CONV_2D,
                            _Op.DEPTHWISE_CONV_2D, _Op.CONV_2D_TRANSPOSE],
                           naive_min_max_quantize.min_max_calibrate, qsv_utils.moving_average_update)
# GPTQ (ref :421-451): FULLY_CONNECTED only; Hessian-collecting calibration + Hessian-merging
# QSV update
_register_weight_algorithm(AlgorithmName.GPTQ, gptq, [_Op.FULLY_CONNECTED], gptq.calibrate,
                           qsv_utils.gptq_and_moving_average_update)

# float casting (ref :166-235): FP16 weights behind a DEQUANTIZE op; its own config check, empty policy
register_op_quant_config_validation_func(AlgorithmName.FLOAT_CASTING, float_casting.check_op_quantization_config)
register_config_check_policy_func(AlgorithmName.FLOAT_CASTING, qtyping.ConfigCheckPolicyDict())
for _op, _fn in ((_Op.FULLY_CONNECTED, float_casting.materialize_fc_conv),
                 (_Op.CONV_2D, float_casting.materialize_fc_conv),
                 (_Op.DEPTHWISE_CONV_2D, float_casting.materialize_fc_conv),
                 (_Op.CONV_2D_TRANSPOSE, float_casting.materialize_conv2d_transpose),
                 (_Op.EMBEDDING_LOOKUP, float_casting.materialize_embedding_lookup)):
  register_quantized_op(AlgorithmName.FLOAT_CASTING, _op, float_casting.init_qsvs,
                        calibration_func=float_casting.calibrate, materialize_func=_fn)

# dequantized weight recovery (ref :203-235): re-derives the integers of fake-quantized weights
register_op_quant_config_validation_func(AlgorithmName.DEQUANTIZED_WEIGHT_RECOVERY,
                                         common_quantize.check_op_quantization_config)
register_config_check_policy_func(AlgorithmName.DEQUANTIZED_WEIGHT_RECOVERY,
                                  default_policy.DEFAULT_CONFIG_CHECK_POLICY)
for _op in (_Op.FULLY_CONNECTED, _Op.CONV_2D, _Op.EMBEDDING_LOOKUP):
  register_quantized_op(
      AlgorithmName.DEQUANTIZED_WEIGHT_RECOVERY, _op, dequantized_weight_recovery.init_qsvs,
      calibration_func=dequantized_weight_recovery.calibrate,
      materialize_func=functools.partial(_MATERIALIZERS[_op],
                                         dequantized_weight_recovery.get_tensor_quant_params))

# OSCAR (ref :453-480): FULLY_CONNECTED only, whole-op materializer, mu2-collecting calibration
register_op_quant_config_validation_func(AlgorithmName.OSCAR, common_quantize.check_op_quantization_config)
register_config_check_policy_func(AlgorithmName.OSCAR, default_policy.DEFAULT_CONFIG_CHECK_POLICY)
register_quantized_op(AlgorithmName.OSCAR, _Op.FULLY_CONNECTED, naive_min_max_quantize.init_qsvs,
                      calibration_func=oscar.calibrate,
                      materialize_func=oscar.materialize_fully_connected,
                      update_qsv_func=qsv_utils.oscar_and_moving_average_update)

# Hadamard rotation (ref :322-383): whole-op materializers, no partial
for _name, _fc, _emb in (
    (AlgorithmName.HADAMARD_ROTATION, hadamard_rotation.materialize_fully_connected_custom_op,
     hadamard_rotation.materialize_embedding_lookup_custom_op),
    (AlgorithmName.DECOMPOSED_HADAMARD_ROTATION,
     hadamard_rotation.materialize_fully_connected_decomposed,
     hadamard_rotation.materialize_embedding_lookup_decomposed)):
  register_op_quant_config_validation_func(_name, common_quantize.check_op_quantization_config)
  register_config_check_policy_func(_name, default_policy.DEFAULT_CONFIG_CHECK_POLICY)
  for _op, _fn in ((_Op.FULLY_CONNECTED, _fc), (_Op.EMBEDDING_LOOKUP, _emb)):
    register_quantized_op(_name, _op, naive_min_max_quantize.init_qsvs,
                          calibration_func=naive_min_max_quantize.min_max_calibrate,
                          materialize_func=_fn)
