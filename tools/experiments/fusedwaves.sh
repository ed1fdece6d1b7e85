cd $GRAFT_REPO_ROOT
for w in 0 1667 1668 1672 1700 1800 1250 1280; do echo "== FUSED_WAVES=$w"; ASPIRE_HIP_FUSED_WAVES=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value %.1f M  one_stream %.1f M  kernel %.1f us  frac %.3f' % (j['value']/1e6, j['one_stream']['value']/1e6, j['roofline']['kernel_ms']*1e3, j['roofline']['frac']))"; done
