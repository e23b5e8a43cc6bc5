"""`python upsnet/upsnet_end2end_test.py ...` -- the reference's entry point (upsnet/upsnet_end2end_test.py:155-290), served by
upsnet_amd/upsnet_end2end_test.py (same loop and timers, process-per-GPU, synthetic inputs)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from upsnet_amd.upsnet_end2end_test import main   # noqa: E402

if __name__ == '__main__':
    main()
