"""Copy what tools/profile_round.sh <tag> left in gpurun_out/ (scratch) into profiles/ (tracked): the tables' inputs as they are, the
bench logs reduced to their JSON line. Usage: python tools/collect_profiles.py <tag>"""
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
for n in ('kernel_stats.txt', 'per_call.txt', 'timeline_serial.txt', 'conv_pmc.json', 'mfma_util_serial.txt', 'layer_table.txt', 'roialign.txt', 'parity.txt',
          'pmc_FETCH_SIZE.txt', 'pmc_WRITE_SIZE.txt', 'timeline_graph_serial.txt', 'c3_conv_pmc.json', 'c3_pmc_FETCH_SIZE.txt', 'c3_pmc_WRITE_SIZE.txt', 'layer_table_c3.txt', 'bf16_kernel_stats.txt', 'bf16_timeline_serial.txt', 'bf16_micro.txt', 'stem_micro.txt', 'winograd36_micro.txt', 'mfma_valu.txt', 'mfma_util_graph.txt', 'conv1x1_ksw.txt', 'conv3x3_ksw.txt', 'winograd36_splitk.txt', 'roi_xcd_order.txt'):
    p = os.path.join(src, '%s_%s' % (tag, n))
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, '%s_%s' % (tag, n)))
for n in ('bench', 'bench_serial', 'bench_bf16', 'bench_bf16x3', 'bench_c3', 'bench_c3_bf16', 'bench_c4', 'bench_ab_deconv_general', 'bench_ab_no_wino36', 'bench_hot', 'dry_run'):
    p = os.path.join(src, '%s_%s.log' % (tag, n))
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith('{')]
        if lines:
            open(os.path.join(dst, '%s_%s.log' % (tag, n)), 'w').write(lines[-1] + '\n')
print(sorted(f for f in os.listdir(dst) if f.startswith(tag + '_')))
