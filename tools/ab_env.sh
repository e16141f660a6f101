#!/bin/bash
# Same-box A/B of one environment switch over the C2 bench legs: tools/ab_env.sh VAR valA valB [rounds]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in $A $B; do
    export $VAR=$v
    python bench.py --no-cpu --no-ar --steps 20 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('$VAR=%s' % os.environ.get('$VAR'), 'train ms %.3f' % d['ms_per_step'], 'glu train us %.2f' % (d['roofline']['avg_launch_ms']*1e3), 'forward ms %.3f' % d['forward_inference']['ms_per_step'], 'glu inference us %.2f' % (d['forward_inference']['roofline']['avg_launch_ms']*1e3), 'loss', d['loss'])"
  done
done
