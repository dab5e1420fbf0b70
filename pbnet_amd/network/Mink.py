"""Import-path mirror of /root/reference/network/Mink.py: `from pbnet_amd.network.Mink import Mink_unet`."""
from .mink_unet import *  # noqa: F401,F403
from .mink_unet import Mink_unet, MinkUNet, SPECS  # noqa: F401
