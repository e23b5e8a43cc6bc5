cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "bf16" 2>&1 | tail -25 > gpurun_out/r08j_pytest.log
tail -25 gpurun_out/r08j_pytest.log
for v in 1 0 1 0; do
  echo "BF16_ACT=$v: $(UPSNET_BF16_ACT=$v timeout 600 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-configs2 --conv-precision bf16 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.2f serial %.3f n_det %d n_inst %d' % (j['value'], j['ms_per_img_serial'], j['config']['n_det'], j['config']['n_inst']))")"
done
