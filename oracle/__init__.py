"""CPU restatement of the kooBH/UNet hot path -- TEST INFRASTRUCTURE ONLY.

This package is the parity checker for the HIP path in ``unet_amd``: plain
``torch.nn`` CPU modules that restate, layer for layer, what the reference's
``UNet`` / ``cRFUNet.UNet20`` / ``ResUNet`` / ``ResUNetOnFreq`` / ``ResUNetOnFreq2`` / ``DSUNet`` /
``UDSS`` / ``DAU`` and the wave-to-wave helpers compute.  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``unet_amd/`` imports it and the product never
routes a forward through it.

Pinning: ``tests/golden/make_golden.py`` (run in the build container,
where the reference checkout is importable) loads identical seeded
state_dicts into the reference modules and into these, and requires
``torch.equal`` outputs in eval mode, in train mode and for gradients.  The
same script writes the golden vectors under ``tests/golden/`` that pin the
oracle where the reference checkout is absent (the GPU box).
"""

from .blocks import (Encoder, Decoder, ResBlock, DSEncoder, DSDecoder, Attractor,
                     AttractEncoder, CRM, CRM_sig, CRM_soft, CM, GConv, TrGConv, TGRUBlock, SEBlock,
                     MultiScaleConvBlock, GlobalChannelLayerNorm, DepthwiseSeparableConv2d, MEA, custom_atan2)
from .nets import UNet, UNet20, ResUNet, ResUNetOnFreq, ResUNetOnFreq2, LSTMBlock, DSUNet
from .tables import unet20_layers, default_unet, dsunet_architecture
from .masking import crf_apply, resunet_mask_apply, crm_apply
from .helper import DSUNet_helper, UDSS_helper, DAU_helper
from .udss import UDSS, DAU, ComplexConv2d, ComplexConvTranspose2d, ComplexBatchNorm2d

__all__ = [
    "Encoder", "Decoder", "ResBlock", "DSEncoder", "DSDecoder", "Attractor",
    "AttractEncoder", "CRM", "CRM_sig", "CRM_soft", "CM", "GConv", "TrGConv", "TGRUBlock", "SEBlock", "MultiScaleConvBlock", "GlobalChannelLayerNorm", "DepthwiseSeparableConv2d", "MEA", "custom_atan2", "UNet", "UNet20", "ResUNet", "ResUNetOnFreq", "ResUNetOnFreq2", "LSTMBlock", "DSUNet",
    "unet20_layers", "default_unet", "dsunet_architecture",
    "crf_apply", "resunet_mask_apply", "crm_apply", "DSUNet_helper",
    "UDSS", "UDSS_helper", "DAU", "DAU_helper", "ComplexConv2d", "ComplexConvTranspose2d", "ComplexBatchNorm2d",
]
