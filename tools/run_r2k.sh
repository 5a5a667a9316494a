for cfg in "5 125 10 5" "10 250 10 10" "5 125 10 5" "10 250 10 10" "20 500 20 0"; do
  set -- $cfg
  echo "== images $1 sub_batch $2 steps $3 warmup $4"
  python bench.py --images $1 --sub_batch $2 --steps $3 --warmup $4 --no_cpu_baseline --no_profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
