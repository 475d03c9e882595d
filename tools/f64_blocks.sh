for bt in 768 512 1024; do
  python bench.py --dtype f64 --no-cpu-baseline --no-large-shop --block-threads $bt 2>/dev/null > gpurun_out/f64_$bt.json
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/f64_$bt.json').read().strip().splitlines()[-1])
print('f64 block $bt', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['bytes_per_launch'])"
done
