"""`from upsnet.models import *` (upsnet/models/__init__.py, used by upsnet_end2end_test.py:42 + eval(config.symbol)())."""
from .resnet_upsnet import resnet_50_upsnet, resnet_101_upsnet

__all__ = ['resnet_50_upsnet', 'resnet_101_upsnet']
