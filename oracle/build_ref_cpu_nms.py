"""TEST INFRASTRUCTURE ONLY -- builds the REFERENCE's own cpu_nms.pyx (cpu_nms, cpu_soft_nms) as a tests-only oracle.

    python oracle/build_ref_cpu_nms.py [/root/reference]

The .pyx is read where it lies under the reference tree, passed through two TYPE-NAME substitutions that numpy >= 1.24 /
Cython 3 force (np.int_t -> np.intp_t, dtype=np.int -> dtype=np.intp: both name the platform's index integer, no arithmetic
changes; `np.float thresh` is left alone -- Cython takes it as a Python float, so `ovr >= thresh` compares in double exactly as
in the reference's build), cythonized with language level 2 (the reference's era) into a scratch directory OUTSIDE the
repository and compiled with gcc; only the extension module lands in oracle/_ref/ (git-ignored, shipped to the GPU box by
gpurun). No reference source is copied into the repository.
"""
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))


def main(ref='/root/reference'):
    src = os.path.join(ref, 'upsnet', 'nms', 'cpu_nms.pyx')
    if not os.path.exists(src):
        print('no reference tree at %s: keeping prebuilt oracle/_ref' % ref)
        return 0
    out_dir = os.path.join(HERE, '_ref')
    os.makedirs(out_dir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='upsnet_ref_cynms_')
    try:
        text = open(src).read()
        text = re.sub(r'np\.int_t', 'np.intp_t', text)
        text = re.sub(r'dtype=np\.int\)', 'dtype=np.intp)', text)
        pyx = os.path.join(tmp, 'upsnet_ref_cpu_nms.pyx')
        with open(pyx, 'w') as f:
            f.write(text)
        c = os.path.join(tmp, 'upsnet_ref_cpu_nms.c')
        subprocess.check_call([sys.executable, '-m', 'cython', '-2', pyx, '-o', c], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        so = os.path.join(out_dir, 'upsnet_ref_cpu_nms' + sysconfig.get_config_var('EXT_SUFFIX'))
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math', '-w',
                               '-I', sysconfig.get_paths()['include'], '-I', numpy.get_include(), c, '-o', so])
        print(so)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == '__main__':
    sys.exit(main(*sys.argv[1:2]))
