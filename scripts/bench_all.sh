# the bench lines of the five networks (cfg4 with the CPU leg): bash scripts/bench_all.sh gpurun_out/tag
set -e
tag=$1
for c in cfg1 cfg2 cfg3 cfg5; do timeout -k 10 400 python bench.py --config $c --steps 10 --warmup 3 > ${tag}_bench_$c.json 2> ${tag}_bench_$c.err; done
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > ${tag}_bench_cfg4.json 2> ${tag}_bench_cfg4.err
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do python -c "
import json; d=json.load(open('${tag}_bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
