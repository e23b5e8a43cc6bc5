cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for v in 192 64 0 192 0; do
  echo "BF16_MIN_WG=$v: $(UPSNET_BF16_MIN_WG=$v timeout 600 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-configs2 --conv-precision bf16 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.2f serial %.3f n_det %d n_inst %d' % (j['value'], j['ms_per_img_serial'], j['config']['n_det'], j['config']['n_inst']))")"
done
