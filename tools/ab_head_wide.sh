for w in 0 1 0 1; do
WAE_HEAD_WIDE=$w python bench.py --mode forward --no-cpu --no-ar --steps 20 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('wide', os.environ.get('WAE_HEAD_WIDE'), 'fwd ms', round(d['ms_per_step'],4), 'loss', d['loss'])"
done
python -m pytest tests/test_gpu_wide.py -q -m gpu 2>&1 | tail -2
