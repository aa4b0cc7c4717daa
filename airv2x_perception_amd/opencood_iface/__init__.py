"""Host-side mirror of the reference's model interface for the Where2Comm-LiDAR hot path.

Module / class names follow ``opencood.models`` so that the reference's name registry
(tools/train_utils.py:288-325 ``create_model``) resolves them through the one-line binding
shown in INTEGRATION.md.
"""
from .airv2x_where2com import Airv2xWhere2com  # noqa: F401
from .airv2x_cobevt import Airv2xCoBEVT  # noqa: F401,E402
from .airv2x_v2xvit import Airv2xV2XVit  # noqa: F401,E402
