for cfg in "wf_pipes=2" "wf_pipes=3" "wf_pipes=3 wf_pool=37748736" "wf_pipes=4 wf_pool=33554432" "wf_pipes=3 wf_waves_per_cu=12" "wf_pipes=4 wf_waves_per_cu=10 wf_pool=33554432"; do
  opts=""; for o in $cfg; do opts="$opts --opt $o"; done
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline $opts 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$cfg ms_per_step', j['ms_per_step'], 'value', j['value'], 'trace avg ms', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])"
done
