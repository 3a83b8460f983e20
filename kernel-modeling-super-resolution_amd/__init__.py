"""kmsr_amd -- MI355X-native degradation-synthesis + SR-CNN hot path (host side).

The directory is named `kernel-modeling-super-resolution_amd`, which is not a valid Python identifier;
`import kmsr_amd` (repo-root shim kmsr_amd.py) loads it under the name `kmsr_amd`.
"""
from . import _lib, dist
from .degradation import (BAND_NAMES, add_noise, add_noise_batch, apply_kernel_degradation,
                          build_noise_pool_from_arrays, degrade_batch, load_kernel, noise_crops, random_crop,
                          uses_fast_path, workspace_for)

from .denoise import denoise_band_float_nlm, estimate_sigma, nl_means
from .drivers import process_files, process_landsat_folder
from .kernelgan import (ConditionEncoder, ConvKFunction, DegradationModel, DynamicMultiBandLinearGenerator,
                        MultiBandLinearGenerator, bulk_extract_kernels, compose_kernels, convk_dgrad, convk_forward, convk_wgrad)
from .train_kernelgan import (crop_patches, kernel_metrics, kernel_regularization_single, kernelgan_multi_step,
                              kernelgan_single_step, load_full_patches, load_patches_from_folder, noise_reg_loss,
                              sample_patches_from_files, train_kernelgan_multi, train_kernelgan_single)
from .patches import apply_water_mask, create_patches
from .moe import ContentAdaptiveDegradation, NoiseEstimator, SelectorNet
from .train_moe import (PatchDiscriminator, kernel_regularization, lsgan_d_loss, lsgan_g_loss, moe_regularizer, save_moe,
                        train_moe)
from .sr_bf16 import (Conv3x3BF16Function, SRNetBF16, conv3x3_bf16, conv3x3_dgrad_bf16, conv3x3_wgrad_bf16,
                      pack_conv3x3_weight_bf16, pack_conv3x3_weight_bf16_dgrad, to_nhwc_bf16)
from .sr import SRNet, conv3x3, pack_conv3x3_weight, pack_conv3x3_weight_wino, psnr

__all__ = ["Conv3x3BF16Function", "conv3x3_dgrad_bf16", "conv3x3_wgrad_bf16", "pack_conv3x3_weight_bf16_dgrad", "ConvKFunction", "convk_dgrad", "convk_forward", "convk_wgrad", "crop_patches", "kernel_metrics",
           "kernel_regularization_single", "kernelgan_multi_step", "kernelgan_single_step", "load_full_patches",
           "load_patches_from_folder", "noise_reg_loss", "sample_patches_from_files", "train_kernelgan_multi",
           "train_kernelgan_single", "PatchDiscriminator", "kernel_regularization", "lsgan_d_loss", "lsgan_g_loss", "moe_regularizer", "save_moe", "train_moe", "denoise_band_float_nlm", "estimate_sigma", "nl_means", "apply_water_mask", "create_patches", "SRNetBF16", "conv3x3_bf16", "pack_conv3x3_weight_bf16", "to_nhwc_bf16", "DegradationModel", "process_files", "process_landsat_folder", "ConditionEncoder", "DynamicMultiBandLinearGenerator", "MultiBandLinearGenerator", "bulk_extract_kernels",
           "compose_kernels", "ContentAdaptiveDegradation", "NoiseEstimator", "SelectorNet", "SRNet", "conv3x3", "pack_conv3x3_weight", "psnr", "_lib", "BAND_NAMES", "add_noise", "add_noise_batch", "apply_kernel_degradation",
           "build_noise_pool_from_arrays", "degrade_batch", "load_kernel", "noise_crops", "random_crop",
           "uses_fast_path", "workspace_for"]
