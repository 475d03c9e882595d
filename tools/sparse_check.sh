python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for cfg in "100 2000000 8" "340 10000 4"; do set -- $cfg
python bench.py --no-cpu-baseline --no-large-shop --cams $1 --timesteps $2 --cams-per-t $3 --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'])"
done
python tools/flake_hunt.py 20 | tail -1
