"""`lib` -- the reference's helper package name (lib/utils/timer.py, lib/utils/data_parallel.py): `lib.utils.X` resolves to
`upsnet_amd.utils.X` (same module objects, see upsnet_amd/_alias.py)."""
from upsnet_amd._alias import install

install('lib.utils', 'upsnet_amd.utils')
