"""GPU: a checkpoint enters through the reference command line (upsnet_end2end_test.py:155-203):
`--cfg <yaml> --weight_path <pth>` -> config.symbol's constructor -> load_state_dict(torch.load(p), resume=True) -> prepare_inference()
-> the process-per-GPU loop. The outputs must equal those of a model built directly from the same tensors."""
import copy

import pytest
import torch

from test_cli_cpu import YAML

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cfg_file(tmp_path):
    from upsnet_amd.config.config import config
    saved = copy.deepcopy(dict(config))
    p = tmp_path / 'exp_r50.yaml'
    p.write_text((YAML % (tmp_path / 'output')).replace('- 64', '- 256').replace('max_size: 128', 'max_size: 512'))
    yield str(p)
    config.clear()
    config.update(saved)


def test_entry_point_runs_a_checkpoint_and_matches_a_directly_built_model(cfg_file, tmp_path):
    from upsnet_amd.config.config import config, update_config
    from upsnet_amd.synthetic import build_unprepared, make_image
    from upsnet_amd.upsnet_end2end_test import main
    update_config(cfg_file)
    src = build_unprepared()                       # the seeded synthetic model as a checkpoint holds it: BN unfolded, reference key names
    sd = src.state_dict()
    assert any(k.endswith('bn3.running_var') for k in sd) and 'fcn_head.score.weight' in sd
    path = str(tmp_path / 'upsnet_resnet_50_cityscapes_12000.pth')
    torch.save({'module.' + k: v for k, v in sd.items()}, path)       # saved from a DataParallel wrapper, as the reference's are
    res = main(['--cfg', cfg_file, '--weight_path', path, '--steps', '3', '--warmup', '1'])
    assert res['world'] == 1 and sorted(res['results']) == [0, 1, 2]
    direct = copy.deepcopy(src).cuda().prepare_inference()
    direct.use_graph = False
    assert config.test.scales[0] == 256 and config.test.max_size == 512
    with torch.no_grad():
        for i in range(3):
            want = direct(make_image(256, 512, seed=i, device='cuda'))
            lab, n_inst = res['results'][i]
            assert torch.equal(lab.to(want['panoptic_outputs'].device).long(), want['panoptic_outputs'][0]), i
            assert n_inst == want['panoptic_cls_inds'].numel()
        for k in ('panoptic_outputs', 'pred_boxes', 'cls_probs', 'mask_probs', 'fcn_outputs', 'cls_inds', 'panoptic_cls_inds'):
            assert torch.equal(res['last_out'][k], want[k]), k
    assert want['cls_inds'].numel() >= 1
