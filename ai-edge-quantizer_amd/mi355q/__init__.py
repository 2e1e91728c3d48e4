"""mi355q -- MI355X-native calibration + requantization behind AI Edge Quantizer's
algorithm_manager / get_tensor_quant_params interface."""
__version__ = "0.1.0"
