for rep in 1 2; do
for v in ${VALUES:-0 224 192 160 128}; do
  FMRI_WW_BLOCKS=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-hbm-rows --no-pmc --no-gate-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('WW_BLOCKS=$v', d['ms_per_step'], d['launch'][:30])"
done; done
