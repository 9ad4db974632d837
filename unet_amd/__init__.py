"""unet_amd -- MI355X-native forward path of the kooBH/UNet model family.

Drop-in ``nn.Module`` classes with the reference's constructors and ``forward``
signatures whose device path is hand-written HIP (libunetmi.so) bound through
``torch.library``.  Importing the package loads the shared library and fails
loudly when it is missing: there is no CPU or eager fallback.
"""

from . import _lib, ops, train  # noqa: F401  (loads libunetmi.so, registers torch.ops.unetmi.*)
from .UNet_m import Encoder, Decoder, ResBlock
from .UNet import UNet, default_unet_blocks
from .cRFUNet import UNet20, UNet10
from .DSUNet import DSUNet, DSUNet_helper
from .ResUNet import ResUNet, ResUNetOnFreq, ResUNetOnFreq2
from .UDSS import UDSS, UDSS_helper
from .DTU import DAU, DAU_helper
from .tables import dsunet_architecture
from .seeding import seeded_fill_

__all__ = ["Encoder", "Decoder", "ResBlock", "UNet", "UNet20", "UNet10", "DSUNet", "DSUNet_helper", "ResUNet", "ResUNetOnFreq", "ResUNetOnFreq2", "UDSS", "UDSS_helper", "DAU", "DAU_helper",
           "default_unet_blocks",
           "dsunet_architecture", "seeded_fill_"]
