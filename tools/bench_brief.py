"""Print the few figures of a bench.py JSON line that matter while tuning (reads the line from stdin)."""
import json
import sys

line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
r = d['roofline']
a = r['all_mfma_convs']
print(sys.argv[1] if len(sys.argv) > 1 else '', 'images/s', d['value'], 'ms/step', d['ms_per_step'],
      '| all convs ms', a['ms_per_step'], 'executed TF', a['executed_tflops'], 'plain TF', a['plain_tflops'],
      '| ACE: TF', r['achieved'], 'avg ms', r['avg_launch_ms'], 'launches', r['launches'], 'exec/dense', r['executed_over_dense'],
      '| interior', (r.get('interior_pass') or {}).get('ms_per_step'))
