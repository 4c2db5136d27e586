O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python bench.py --steps 100 --warmup 20 > $O/bench_stage1.json 2> $O/bench_stage1.err; cut -c1-160 $O/bench_stage1.json
for w in stage2 dual1 stage3_px128; do timeout -k 10 300 python bench.py --workload $w --steps 100 --warmup 20 > $O/bench_$w.json 2> $O/bench_$w.err; cut -c1-120 $O/bench_$w.json; done
timeout -k 10 400 bash tools/prof_stats.sh final/prof --no-hbm-rows --no-pmc --no-gate-pass > /dev/null 2>&1
timeout -k 10 500 bash tools/pmc_passes.sh final/pmc > $O/pmc.log 2>&1
timeout -k 10 600 bash tools/bench_median5.sh > $O/median5.json 2> /dev/null; cat $O/median5.json
(python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-hbm-rows --no-pmc --no-gate-pass 2>/dev/null; FMRI_FORCE_DIST=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-hbm-rows --no-pmc --no-gate-pass 2>/dev/null) | python -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(json.dumps({'plain_ms_per_step': r[0]['ms_per_step'], 'force_dist_ms_per_step': r[1]['ms_per_step'], 'overhead': round(r[1]['ms_per_step']/r[0]['ms_per_step']-1,4), 'launch': [r[0]['launch'], r[1]['launch']]}))" > $O/force_dist.json; cat $O/force_dist.json
ls $O $O/prof $O/pmc
