"""`UPSNET_*` environment knobs: which ones exist, which ones are set.

The tuning / diagnosis switches of this package are read from the environment at import or first use (models/hipconv.py, ops.py,
models/resnet_upsnet.py, csrc/*.hip via getenv). None of them is set in a default run, so every `UPSNET_*` variable present in the
environment is a deviation from the benchmarked configuration: bench.py prints them into `config.knobs`, and a name that no source
file reads (a typo, a knob of an older build) is an error instead of a silently different kernel mix."""
import os
import re

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = re.compile(r'UPSNET_[A-Z0-9_]+')


def known():
    """Every UPSNET_* name some source file of the package (or bench.py) mentions."""
    names = set()
    files = [os.path.join(os.path.dirname(_ROOT), 'bench.py')]
    for d, _, fs in os.walk(_ROOT):
        files += [os.path.join(d, f) for f in fs if f.endswith(('.py', '.hip', '.cpp', '.h'))]
    for f in files:
        try:
            with open(f, errors='ignore') as fh:
                names.update(_NAME.findall(fh.read()))
        except OSError:
            pass
    return names


def active(env=None):
    """{name: value} of the UPSNET_* variables set in the environment (= the non-default knobs of this run)."""
    env = os.environ if env is None else env
    return {k: env[k] for k in sorted(env) if k.startswith('UPSNET_')}


def check(env=None):
    """active(), after making sure every set knob is one the sources read; raises ValueError on an unknown name."""
    act = active(env)
    unknown = sorted(set(act) - known())
    if unknown:
        raise ValueError('unknown UPSNET_* environment variable(s) %s -- no source file reads them (typo?)' % ', '.join(unknown))
    return act
