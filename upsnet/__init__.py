"""`upsnet` -- the reference's package name, served by the MI355X-native implementation in `upsnet_amd`.

`import upsnet.operators.modules.deform_conv`, `from upsnet.nms.nms import gpu_nms_wrapper`, `from upsnet.models import *`,
`from upsnet.config.config import config` ... resolve to the modules of the same relative path under `upsnet_amd` (the same module
objects, see upsnet_amd/_alias.py), so that a caller written against the reference's tree (upsnet/upsnet_end2end_test.py:31-47 and
the model files' own imports) runs on this implementation with its import lines unchanged.
"""
import upsnet_amd
from upsnet_amd._alias import install

install('upsnet', 'upsnet_amd')
__path__ = []   # no files of its own besides upsnet_end2end_test.py's launcher: every submodule comes from the finder
