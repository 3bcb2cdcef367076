for pool in 0 16777216 25165824; do
  for cfg in "--steps 10 --warmup 2" "--spp 1 --steps 320 --warmup 32" "--width 3840 --height 2160 --spp 256 --steps 2 --warmup 1"; do
    python bench.py $cfg --no-cpu-baseline --opt wf_pool=$pool 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('pool=$pool [$cfg] ms_per_step', j['ms_per_step'], 'value', j['value'], 'frac', j['roofline']['frac'], 'whole', j['roofline']['whole_pass_frac'])"
  done
done
